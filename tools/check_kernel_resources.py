#!/usr/bin/env python3
"""Post-build check of the solve kernels' register budgets (the kernel-descriptor notes of the built object): the instantiations
that batches beyond one solve per SIMD run on must leave room for TWO wavefronts per SIMD — at most 256 vector + accumulator
registers — and the production instantiations must not spill vector registers to scratch.  A refactoring once pushed the
one-wavefront sampled kernel to 257 + 1 registers unnoticed: −7 % at B = 4096, −20 % at B = 8192.

    python tools/check_kernel_resources.py [path/to/cilqr_solve.o]      exit code 1 and a listing on any violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes(obj):
    """{kernel name: {vgpr, agpr, vgpr_spill, sgpr_spill, scratch}} of the device code object inside `obj`."""
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "o.o")
        with open(obj, "rb") as f, open(tmp, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [os.path.join(d, n) for n in os.listdir(d) if "amdgcn" in n]
        if not dev:
            raise SystemExit("no device code object in %s" % obj)
        text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev[0]], check=True, capture_output=True, text=True).stdout
    out = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", text, re.S):
        blk = m.group(0)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
        out[re.search(r"\.name:\s+(\S+)", blk).group(1)] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), vgpr_spill=g("vgpr_spill_count"),
                                                               sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"))
    return out


def check(notes):
    """Violations among the solve kernels.  Mangled names: cilqr_solve_kernelIL b<DIAG> E Li<TAB> E L b<GENERAL> E L b<UNC> E."""
    bad = []
    for name, r in sorted(notes.items()):
        m = re.search(r"cilqr_solve_kernelILb(\d)ELi(\d)ELb(\d)ELb(\d)E", name)
        split = re.search(r"cilqr_solve_split_kernelILi(\d)ELb(\d)ELb(\d)E", name)
        share = re.search(r"cilqr_solve_share_kernelILi(\d)ELb\dELb(\d)ELb(\d)E", name)
        if m:
            diag, tab, general, unc = (int(v) for v in m.groups())
            if not unc and r["vgpr"] + r["agpr"] > 256:
                bad.append((name, r, "more than 256 vector registers: one wavefront per SIMD only"))
            if not diag and not general and not unc and tab != 2 and r["vgpr_spill"]:
                bad.append((name, r, "a production instantiation spills vector registers"))
        elif split or share:
            # (three wavefronts per solve: three per SIMD at one solve per SIMD; with a map set the kernel is built for two per SIMD)
            budget = 168 if share and int(share.group(1)) == 3 and not int(share.group(3)) else 256
            if r["vgpr"] + r["agpr"] > budget:
                bad.append((name, r, "more than %d vector registers: the workgroups of a CU halve" % budget))
            if not int((split or share).group(2)) and r["vgpr_spill"]:
                bad.append((name, r, "a production instantiation spills vector registers"))
    return bad


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd", "build", "cilqr_solve.o")
    notes = kernel_notes(obj)
    bad = check(notes)
    n = sum(1 for k in notes if "cilqr_solve" in k)
    print("%d solve kernels checked in %s: %d violation(s)" % (n, os.path.relpath(obj, ROOT), len(bad)))
    for name, r, why in bad:
        print("  %s\n    %s\n    %s" % (name, r, why))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
