#!/usr/bin/env python3
"""Dependent-memory-round-trip view of the gfx950 ISA of one source file (no GPU needed).

A short kernel's duration is, to first order, the number of memory round trips the compiler left DEPENDENT on each
other.  This tool compiles a .hip file with `hipcc -S` and prints, per kernel, the order of

    L  global/buffer/scratch load      S  scalar load (s_load / s_buffer_load)     A  returning atomic
    W  s_waitcnt vmcnt(n) that waits for at least one outstanding vector load (n below the loads in flight)
    w  s_waitcnt lgkmcnt(0) that waits for a scalar load          B  s_barrier          |  branch target / loop head

so that "LWLWLW" (a chain: each load waits for the previous one) stands out against "LLLLW" (one round trip).  Runs of the
same letter are printed with a count (L8 = eight loads requested back to back).

    python tools/isa_chains.py gcc_amd/csrc/encoder.hip [kernel-name-substring ...]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

HIPCC = "/opt/rocm/bin/hipcc"
ROOT = Path(__file__).resolve().parent.parent


def isa_of(src: Path, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", str(src.parent),
               *extra, "-o", str(out), str(src)]
        subprocess.run(cmd, check=True)
        return out.read_text()


def kernels(text):
    """yield (mangled name, list of instruction lines) for every amdgpu kernel in the listing"""
    names = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", text))
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and m.group(1) in names:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):            # (not the first s_endpgm: early exits end the program too)
            yield cur, body
            cur = None
            continue
        body.append(s)


def demangle(name):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            if out:
                return out
        except OSError:
            pass
    return name


def chain(body):
    out = []
    in_flight = 0            # vector loads requested and not yet waited for
    s_in_flight = 0
    for s in body:
        op = s.split()[0] if s else ""
        if re.match(r"^\.?LBB\d+_\d+:", s):
            out.append("|")
        elif re.match(r"^(global|buffer|scratch|flat)_load", op):
            out.append("L")
            in_flight += 1
        elif re.match(r"^(global|buffer|flat)_atomic", op) and (" glc" in s or " sc0" in s):
            out.append("A")
            in_flight += 1
        elif re.match(r"^s_(buffer_)?load", op):
            out.append("S")
            s_in_flight += 1
        elif op == "s_barrier":
            out.append("B")
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m and int(m.group(1)) < in_flight:
                out.append("W")
                in_flight = int(m.group(1))
            if "lgkmcnt(0)" in s and s_in_flight:
                out.append("w")
                s_in_flight = 0
    # run-length
    res, i = [], 0
    while i < len(out):
        j = i
        while j < len(out) and out[j] == out[i]:
            j += 1
        res.append(out[i] if j - i == 1 or out[i] == "|" else f"{out[i]}{j - i}")
        i = j
    return "".join(res)


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src = Path(sys.argv[1])
    if not src.is_absolute():
        src = ROOT / src
    want = sys.argv[2:]
    text = isa_of(src)
    for name, body in kernels(text):
        pretty = demangle(name)
        if want and not any(w in pretty for w in want):
            continue
        c = chain(body)
        waits = c.count("W")
        m = re.search(r"\d+([a-z0-9_]+_kernel(?:IL[a-z0-9_]+E)?)", name)
        print(f"{m.group(1) if m else pretty}: {len(body)} lines, {waits} vector-load waits")
        print("   " + c)


if __name__ == "__main__":
    main()
