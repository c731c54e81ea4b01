#!/usr/bin/env python3
"""Post-build ISA check of the scalar-path forward pass: in every solve kernel that contains s_load_dwordx16 (the
forward-pass records, forward_smem in cilqr_solve.hip) no instruction may read or write a scalar register of a loaded range
between the load and the next s_waitcnt lgkmcnt(0) — the loads are inline assembly the compiler does not see, so a copy, spill or
re-materialisation it schedules in between would silently use registers whose load has not landed.

    python tools/check_smem_hazard.py [path/to/cilqr_solve.o]      exit code 1 and a listing on any violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "o.o")
        with open(obj, "rb") as f, open(tmp, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", tmp], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [os.path.join(d, n) for n in os.listdir(d) if "amdgcn" in n]
        if not dev:
            raise SystemExit("no device code object in %s" % obj)
        return subprocess.run([OBJDUMP, "-d", dev[0]], check=True, capture_output=True, text=True).stdout


def sregs(text):
    """Scalar registers named in an operand string: s5 and s[4:7] → {4, 5, 6, 7}."""
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return out


def parse(asm):
    """{kernel: [(offset, instruction text, branch target offset or None)]}"""
    kernels, cur, base = {}, None, 0
    for line in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            base, cur = int(m.group(1), 16), kernels.setdefault(m.group(2), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        code, _, comment = line.partition("//")
        ins = code.strip()
        a = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not ins or not a:
            continue
        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>\s*$", comment)
        cur.append((int(a.group(1), 16) - base, ins, int(t.group(1), 16) if t else None))
    return kernels


def check(asm):
    """Follows the control flow from every s_load_dwordx16 to the s_waitcnt lgkmcnt(0) that ends its flight."""
    violations, loads = [], 0
    for name, code in parse(asm).items():
        index = {off: i for i, (off, _, _) in enumerate(code)}
        for i0, (_, ins0, _) in enumerate(code):
            if not ins0.startswith("s_load_dwordx16"):
                continue
            loads += 1
            dst = re.match(r"s_load_dwordx16\s+s\[(\d+):(\d+)\]", ins0)
            regs = set(range(int(dst.group(1)), int(dst.group(2)) + 1))
            seen, work = set(), [i0 + 1]
            while work:
                i = work.pop()
                while i < len(code) and i not in seen:
                    seen.add(i)
                    _, ins, target = code[i]
                    op = ins.split()[0]
                    if op == "s_waitcnt" and "lgkmcnt(0)" in ins:
                        break
                    if op == "s_endpgm":
                        violations.append((name, ins0, "reaches s_endpgm with the load in flight"))
                        break
                    if op.startswith("s_load_dwordx16"):
                        again = re.match(r"s_load_dwordx16\s+s\[(\d+):(\d+)\]", ins)
                        if regs & set(range(int(again.group(1)), int(again.group(2)) + 1)):
                            violations.append((name, ins0, "its registers are the target of another load before the wait: " + ins))
                        if sregs(ins.split(",", 1)[1]) & regs:
                            violations.append((name, ins0, "its registers address another load before the wait: " + ins))
                    elif sregs(ins) & regs:
                        violations.append((name, ins0, "touched between the load and the wait by: " + ins))
                    if op == "s_branch" and target is not None:
                        i = index.get(target, len(code))
                        continue
                    if op.startswith("s_cbranch") and target is not None and target in index:
                        work.append(index[target])
                    i += 1
    return violations, loads


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd", "build", "cilqr_solve.o")
    violations, loads = check(disassemble(obj))
    print("%d s_load_dwordx16 checked in %s: %d violation(s)" % (loads, os.path.relpath(obj, ROOT), len(violations)))
    for name, ins, why in violations:
        print("  %s\n    %s\n    %s" % (name, ins, why))
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())
