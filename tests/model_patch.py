"""Test helper: walks the model container (layout: tools/make_synth_model.py, /root/reference/convert.py:202-322) and
returns the byte offsets of every hparam and tensor-record field, so that tests can damage single fields of a copy."""
import struct

MAGIC = 0x67676D6C
HP_NAMES = ("n_layer", "n_head", "n_embd", "block_size", "bias", "n_in_vocab", "n_out_vocab", "n_lm_heads", "n_wtes", "ftype")
TYPE_BYTES = {0: (4, 1), 1: (2, 1), 2: (18, 32), 3: (20, 32), 6: (22, 32), 7: (24, 32), 8: (34, 32)}     # ggml_type -> (bytes, per elements)


def _record(buf, pos):
    n_dims, name_len, ttype = struct.unpack_from("<iii", buf, pos)
    dims_off = pos + 12
    dims = struct.unpack_from("<%di" % n_dims, buf, dims_off)
    name_off = dims_off + 4 * n_dims
    name = bytes(buf[name_off:name_off + name_len]).decode()
    n = 1
    for d in dims:
        n *= d
    b, per = TYPE_BYTES[ttype]
    data_off = name_off + name_len
    end = data_off + n // per * b
    return name, {"rec": pos, "n_dims_off": pos, "ttype_off": pos + 8, "dims_off": dims_off, "dims": dims, "data_off": data_off, "end": end}, end


def walk(buf):
    """-> {"gpt": [{"hp_off": {name: offset}, "hp": {...}, "tensors": {name: info}} x 3], "codec_hp_off": off, "codec": {name: info}}"""
    pos = 0
    assert struct.unpack_from("<I", buf, pos)[0] == MAGIC
    pos += 4
    n_vocab = struct.unpack_from("<i", buf, pos)[0]
    pos += 4
    for _ in range(n_vocab):
        ln = struct.unpack_from("<I", buf, pos)[0]
        pos += 4 + ln
    out = {"gpt": [], "codec": {}}
    for _ in range(3):
        hp_off = {n: pos + 4 * i for i, n in enumerate(HP_NAMES)}
        hp = dict(zip(HP_NAMES, struct.unpack_from("<10i", buf, pos)))
        pos += 40
        n_t = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        tensors = {}
        for _ in range(n_t):
            name, info, pos = _record(buf, pos)
            tensors[name] = info
        out["gpt"].append({"hp_off": hp_off, "hp": hp, "tensors": tensors})
    assert struct.unpack_from("<I", buf, pos)[0] == MAGIC
    out["codec_hp_off"] = pos + 4
    pos += 4 + 36
    while pos < len(buf):
        name, info, pos = _record(buf, pos)
        out["codec"][name] = info
    return out


def poke_i32(buf, off, value):
    struct.pack_into("<i", buf, off, value)
