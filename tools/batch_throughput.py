"""Throughput of in-engine batching (bark_hip_generate_batch) on ONE GPU: B utterances in lock step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
n_sem = int(sys.argv[1]) if len(sys.argv) > 1 else 256
preset = sys.argv[2] if len(sys.argv) > 2 else "small"
prompts = bench.synth_prompts(64)
for B in (1, 2, 4, 8, 16, 32, 64):
    ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=n_sem), 0)
    ctx.generate_batch(prompts[:B])                      # warm-up (graph capture, allocations)
    t0 = time.perf_counter()
    res = ctx.generate_batch(prompts[B:2 * B] if 2 * B <= 64 else prompts[:B])
    dt = time.perf_counter() - t0
    audio = sum(len(r["pcm"]) for r in res) / 24000.0
    st = ctx.stats()
    print(f"B={B:2d} wall={dt * 1e3:8.1f} ms  prompts/s={B / dt:6.2f}  audio-s/s={audio / dt:7.2f}  semantic {st['t_semantic_us'] / 1e3:.0f} ms coarse {st['t_coarse_us'] / 1e3:.0f} ms fine {st['t_fine_us'] / 1e3:.0f} ms codec {st['t_codec_us'] / 1e3:.0f} ms", flush=True)
    if B == 64:
        # the ragged form of the 64-prompt job (bench.ragged_caps: step caps 64..256 by prompt length) on the same context
        caps = bench.ragged_caps(prompts)
        reqs = [ctx.request_params(n_steps_text_encoder=caps[i]) for i in range(64)]
        ctx.generate_batch(prompts, params=reqs)
        t0 = time.perf_counter(); res = ctx.generate_batch(prompts, params=reqs); dt = time.perf_counter() - t0
        st = ctx.stats()
        print(f"ragged 64 (caps 64..256) wall={dt * 1e3:8.1f} ms  prompts/s={64 / dt:6.2f}  audio-s/s={sum(len(r['pcm']) for r in res) / 24000.0 / dt:7.2f}  semantic {st['t_semantic_us'] / 1e3:.0f} ms coarse {st['t_coarse_us'] / 1e3:.0f} ms fine {st['t_fine_us'] / 1e3:.0f} ms codec {st['t_codec_us'] / 1e3:.0f} ms", flush=True)
    ctx.free()
