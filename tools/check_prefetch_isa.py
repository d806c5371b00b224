#!/usr/bin/env python3
"""check_prefetch_isa.py FILE.s ... - static check of the opt-in weight-prefetch kernels (NextWeights, device_utils.h) in a device assembly
(hipcc -S --cuda-device-only).  The prefetch requests are loads nobody waits for; that is only safe if the register they return into is
never given to another value while a request may be in flight.  For every kernel that contains a request (marked `; NWPF`) the check
demands: (1) all requests of the kernel write the SAME VGPR; (2) apart from the requests, the `v_mov` that initialises it and the
`; NWPF hold` marker that ends its live range, NO instruction of the kernel names that register (alone or inside a register range);
(3) no scratch (a spilled sink would be reloaded over an in-flight request).  Exit status 1 on a violation."""
import re
import sys


def kernel_bodies(path):
    name, body = None, []
    for line in open(path, errors="replace"):
        s = line.rstrip("\n")
        m = re.match(r"^(_Z[_A-Za-z0-9$.]+):", s)
        if m and not m.group(1).endswith(".kd"):
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is not None:
            if s.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(s.strip())
    if name:
        yield name, body


def names_register(instr, reg):
    code = instr.split(";")[0]
    if re.search(r"\bv%d\b" % reg, code):
        return True
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        if int(a) <= reg <= int(b):
            return True
    return False


def check(path):
    """Per kernel and per sink register R: the region from the first request into R to the next `NWPF hold` naming R (text order; the kernels
    have no backward branch across a request site other than the request loops themselves) must name R in requests only."""
    bad = 0
    seen = 0
    for name, body in kernel_bodies(path):
        code = [l for l in body if l and not l.startswith(".") and not (l.startswith(";") and "NWPF" not in l)]
        req_idx = [i for i, l in enumerate(code) if "NWPF" in l and l.startswith("global_load_dword")]
        if not req_idx:
            continue
        seen += 1
        regs = sorted({int(re.match(r"global_load_dword v(\d+),", code[i]).group(1)) for i in req_idx})
        ok = True
        notes = []
        for reg in regs:
            first = next(i for i in req_idx if re.match(r"global_load_dword v%d," % reg, code[i]))
            hold = next((i for i in range(first, len(code)) if "NWPF hold" in code[i] and re.search(r"\bv%d\b" % reg, code[i])), None)
            if hold is None:
                print(f"FAIL {name}: no hold marker for v{reg} behind its first request"); ok = False; continue
            later = [i for i in req_idx if i > hold and re.match(r"global_load_dword v%d," % reg, code[i])]
            if later:
                print(f"FAIL {name}: a request into v{reg} behind its hold marker"); ok = False; continue
            # a `v_mov_b32 vR, 0` inside the region is the sink's own initial value re-materialised on a path that joins behind a request site
            # (the two sites of a kernel are alternatives): a write nobody reads - any READER of vR would be listed here and fail the check
            rest = [code[i] for i in range(first, hold) if "NWPF" not in code[i] and names_register(code[i], reg)
                    and not re.match(r"v_mov_b32(_e32)? v%d, 0$" % reg, code[i].split(";")[0].strip())]
            if rest:
                print(f"FAIL {name}: v{reg} (prefetch sink) is named between its first request and its hold marker by: {rest[:4]}"); ok = False; continue
            tail = [l for l in code[hold + 1:hold + 40]]
            end = next((k for k, l in enumerate(tail) if l.startswith("s_endpgm")), None)
            notes.append(f"v{reg}: {sum(1 for i in req_idx if first <= i < hold)} request site(s), hold {hold - first} instructions later, s_endpgm {end if end is not None else '>40'} behind the hold")
        if ok:
            print(f"ok   {name}: " + "; ".join(notes))
        else:
            bad += 1
    return bad, seen


def main():
    total_bad = total_seen = 0
    for p in sys.argv[1:]:
        b, s = check(p)
        total_bad += b; total_seen += s
    print(f"{total_seen} prefetching kernels checked, {total_bad} violations")
    sys.exit(1 if total_bad or not total_seen else 0)


if __name__ == "__main__":
    main()
