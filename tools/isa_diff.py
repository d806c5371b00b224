#!/usr/bin/env python3
"""isa_diff.py OLD.s NEW.s [--map 'old_regex=>new_text' ...] - compares the gfx950 instruction streams of two device assemblies
(hipcc -S --cuda-device-only) kernel by kernel.  Used to show that a change which adds template variants leaves the default
instantiations' machine code untouched (no GPU needed).  Kernel names are demangled; --map rewrites OLD names to the NEW spelling
(e.g. 'gemv_kernel<(\\d+)>=>gemv_kernel<\\1, false>').  Prints per kernel: identical / differs (first differing line) / only in one file,
plus VGPR / SGPR / scratch of the new one."""
import re
import subprocess
import sys


def kernels(path):
    out, name, body, meta = {}, None, [], {}
    for line in open(path, errors="replace"):
        s = line.strip()
        m = re.match(r"^([_A-Za-z0-9$.]+):\s*(;.*)?$", s)
        if m and m.group(1).startswith("_Z") and not m.group(1).endswith(".kd"):
            name, body = m.group(1), []
            out[name] = (body, {})
            continue
        m2 = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|NumAgprs|Occupancy): (\d+)", s)
        if m2 and out:
            out[list(out)[-1]][1][m2.group(1)] = int(m2.group(2))
            continue
        if name is None:
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".Lfunc_end"):
            if s.startswith(".Lfunc_end"):
                name = None
            continue
        if not s or s.startswith(";") or s.startswith("."):
            m2 = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|NumAgprs|Occupancy): (\d+)", s)
            if m2 and out:
                last = list(out)[-1]
                out[last][1][m2.group(1)] = int(m2.group(2))
            continue
        body.append(re.sub(r"\s*;.*$", "", s))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def main():
    old, new = kernels(sys.argv[1]), kernels(sys.argv[2])
    maps = []
    args = sys.argv[3:]
    while args:
        if args[0] == "--map":
            a, b = args[1].split("=>"); maps.append((re.compile(a), b)); args = args[2:]
        else:
            raise SystemExit("unknown argument " + args[0])
    dold, dnew = demangle(list(old)), demangle(list(new))
    def short(n):
        if n.startswith("_Z"):
            return n                     # the system's c++filt does not know _Float16 (DF16_): such names stay mangled, --map then works on the mangling
        return re.sub(r"\(.*$", "", n.replace("void barkhip::", "").replace("barkhip::", ""))
    o2 = {}
    for k, v in dold.items():
        n = short(v)
        for rx, rep in maps:
            n = rx.sub(rep, n)
        o2[n] = k
    n2 = {short(v): k for k, v in dnew.items()}
    same = diff = 0
    for n in sorted(set(o2) | set(n2)):
        if n not in o2:
            meta = new[n2[n]][1]
            print(f"new only   {n}  {meta}")
        elif n not in n2:
            print(f"old only   {n}")
        else:
            a, b = old[o2[n]][0], new[n2[n]][0]
            # labels carry function-local numbering: normalise
            na = [re.sub(r"\.LBB\d+_", ".LBB_", x) for x in a]
            nb = [re.sub(r"\.LBB\d+_", ".LBB_", x) for x in b]
            # a by-value argument struct that grew moves the kernel-argument offsets of what follows it: s_load offsets are not code
            ka = [re.sub(r"0x[0-9a-f]+$", "OFF", x) if x.startswith("s_load_") else x for x in na]
            kb = [re.sub(r"0x[0-9a-f]+$", "OFF", x) if x.startswith("s_load_") else x for x in nb]
            if na == nb:
                same += 1
            elif ka == kb:
                same += 1
                print(f"same code  {n}: identical up to kernel-argument offsets")
            else:
                diff += 1
                i = next((i for i, (x, y) in enumerate(zip(na, nb)) if x != y), min(len(na), len(nb)))
                print(f"DIFFERS    {n}: {len(na)} vs {len(nb)} instructions, first difference at {i}: {na[i] if i < len(na) else None!r} / {nb[i] if i < len(nb) else None!r}")
    print(f"{same} kernels identical, {diff} differ")


if __name__ == "__main__":
    main()
