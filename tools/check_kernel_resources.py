#!/usr/bin/env python
"""Build-time guard (yolo_amd/csrc/Makefile): fail when a kernel that gates LDS reads on hand-counted
`s_waitcnt vmcnt(N)` has scratch or spills.

conv3x3_c32, conv3x3s2_c64, conv1x1_regw, resblock64 / resblock128 and the ping-pong implicit GEMM count their own
stores / LDS-DMA requests in the tile loop and wait with vmcnt(N), N > 0.  (resblock_kernel<256> and the stem drain with
vmcnt(0) or leave the count to the compiler: a spill there costs time, not correctness.)  That is only correct while the compiler emits no
other vector-memory instruction there: one scratch spill (buffer_store / buffer_load count in vmcnt) after a compiler or
flag change would let a wave read a patch before its LDS-DMA has landed -- silently wrong, not a fault.  Input: the
compiler's -Rpass-analysis=kernel-resource-usage remarks of one translation unit."""
import re
import sys

GUARDED = ("conv3x3_c32_kernel", "conv3x3s2_c64_kernel", "conv1x1_regw_kernel", "resblock64_kernel", "resblock128_kernel",
           "conv_pp_kernel", "conv_pp_split_kernel")


def main(path):
    text = open(path, errors="replace").read()
    bad, seen = [], 0
    for m in re.finditer(r"remark: Function Name: (\S+)", text):
        name = m.group(1)
        if not any(g in name for g in GUARDED):
            continue
        seen += 1
        block = text[m.end():m.end() + 4000]
        nxt = block.find("remark: Function Name:")
        block = block if nxt < 0 else block[:nxt]
        vals = {k: int(v) for k, v in re.findall(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", block)}
        # (SGPRs spilled to VGPR lanes -- v_writelane / v_readlane, ScratchSize 0 -- issue no vector-memory instruction)
        if vals.get("ScratchSize [bytes/lane]", 0) or vals.get("VGPRs Spill", 0):
            bad.append((name, vals))
    for name, vals in bad:
        sys.stderr.write("check_kernel_resources: %s has scratch / spills %r: its counted vmcnt(N) waits are no longer "
                         "exact\n" % (name, vals))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
