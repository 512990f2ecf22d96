#!/usr/bin/env python
"""What hipcc made of the kernels, without a GPU: per kernel the register / scratch numbers and the tell-tales of a
bad allocation (SGPRs spilled lane by lane: v_readlane / v_writelane; scratch loads and stores), and, for one kernel,
its basic blocks with the instruction mix of each (where the MFMAs, LDS reads, LDS-DMA loads and waits sit).

    python tools/asm_scan.py                         # every .hip under yolo_amd/csrc: one line per kernel
    python tools/asm_scan.py conv_igemm 'conv_igemm_kernelILb1ELi256ELi256'     # blocks of the kernels matching

Findings this found in round 1: the fused first-layer kernel kept 432 scalar weights live and spilled them to lanes
(1 800 v_readlane per thread); the 256x256 tile's second half-step reuses one register quad for its four weight
fragments; every wait on an LDS read after an LDS-DMA load is lgkmcnt(0).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]


def assemble(stem, extra=()):
    out = os.path.join(tempfile.gettempdir(), "y3_%s.s" % stem)
    subprocess.run(["hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, stem + ".hip"), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(asm):
    """-> [(mangled name, body text, resource comment block)]"""
    res = []
    for m in re.finditer(r"^(_Z\S+):\s*; @", asm, re.M):
        end = asm.find(".amdhsa_kernel", m.start())
        if end < 0:
            continue
        info = asm.find("; NumVgprs", end)
        res.append((m.group(1), asm[m.start():end], asm[info:info + 400] if info > 0 else ""))
    return res


def field(info, key):
    m = re.search(r"; %s: (\d+)" % key, info)
    return int(m.group(1)) if m else -1


def summary(stem, extra):
    for name, body, info in kernels(assemble(stem, extra)):
        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:70]
        print("%-12s %-70s vgpr %3d agpr %3d scratch %4d B occupancy %d | readlane %4d writelane %4d scratch ops %3d mfma %4d" % (
            stem, short, field(info, "NumVgprs"), field(info, "NumAgprs"), field(info, "ScratchSize"), field(info, "Occupancy"),
            body.count("v_readlane_b32"), body.count("v_writelane_b32"), len(re.findall(r"\bscratch_(load|store)", body)),
            body.count("v_mfma")))


def blocks(stem, pattern, extra):
    for name, body, info in kernels(assemble(stem, extra)):
        if pattern not in name:
            continue
        print(name)
        cur, ins = "entry", []
        out = []
        for line in body.split("\n"):
            t = line.strip()
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                out.append((cur, ins))
                cur, ins = m.group(1), []
            elif t and not t.startswith((";", ".")) and line.startswith("\t"):
                ins.append(t)
        out.append((cur, ins))
        for label, ins in out:
            c = collections.Counter(x.split()[0] for x in ins)
            n = lambda pred: sum(v for k, v in c.items() if pred(k))
            if not ins:
                continue
            print("  %-12s %4d instr  mfma %3d  ds_read %3d  lds-dma %2d  valu %4d  salu %4d  waitcnt %2d  barrier %d  lane-spill %3d  -> %s" % (
                label, len(ins), n(lambda k: "mfma" in k), n(lambda k: k.startswith("ds_read")),
                n(lambda k: k.startswith("global_load_lds") or (k.startswith("buffer_load") and k.endswith("lds"))),
                n(lambda k: k.startswith("v_") and "mfma" not in k), n(lambda k: k.startswith("s_")), c.get("s_waitcnt", 0),
                c.get("s_barrier", 0), n(lambda k: k in ("v_readlane_b32", "v_writelane_b32")),
                " ".join(x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch")))))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    if len(args) >= 2:
        blocks(args[0], args[1], extra)
    else:
        for f in sorted(os.listdir(CSRC)):
            if f.endswith(".hip") and (not args or f[:-4] == args[0]):
                summary(f[:-4], extra)
