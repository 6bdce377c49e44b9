#!/usr/bin/env python3
"""Per-kernel ISA statistics of a hipcc -S --cuda-device-only listing: MFMA count, scratch (spill) accesses and where they sit
relative to the MFMAs, AGPR<->VGPR moves, SGPR spill lane operations, s_nop.  usage: isa_stats.py file.s [name-substring]"""
import bisect
import re
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_wf4x_kernel"
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+:\s", l) and pat in l]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        body = lines[st:end]
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        scr = [i for i, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
        acc = sum(("v_accvgpr_read" in l or "v_accvgpr_write" in l) for l in body)
        lane = sum(("v_writelane" in l or "v_readlane" in l) for l in body)
        nop = sum(l.strip().startswith("s_nop") for l in body)
        wait = sum(l.strip().startswith("s_waitcnt") for l in body)
        print(f"{body[0].split(':')[0][:70]}: {len(body)} lines, {len(mf)} mfma, {len(scr)} scratch, {acc} accvgpr moves, {lane} lane ops, {nop} s_nop, {wait} waitcnt")
        if scr:
            print("   scratch accesses behind MFMA #:", [bisect.bisect(mf, i) for i in scr][:80])


if __name__ == "__main__":
    main()
