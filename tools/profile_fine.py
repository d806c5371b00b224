"""A few fine forward passes of bark-small (f16, or q4_0 with `q4_0` as argument): the workload of the MFMA PMC profile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
path = ensure_model("small", 0)
if len(sys.argv) > 1 and sys.argv[1] != "f16":
    ftype = {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}[sys.argv[1]]
    q = path[:-4] + "_%s.bin" % sys.argv[1]
    if not os.path.exists(q):
        assert pkg.load_library().bark_model_quantize(path.encode(), q.encode(), ftype)
    path = q
ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0), 0)
Z = int(os.environ.get("FINE_WINDOWS", "1"))
print("fine pass", ctx.time_fine_pass(3, Z))
ctx.free()
