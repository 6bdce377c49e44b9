#!/usr/bin/env python3
"""Find the store-data hazard measured in round 4 (conv3x3_wf4.h, GENERAL epilogue) in gfx950 code: a store of more than 8 bytes
whose data VGPRs are overwritten by a VECTOR-ALU instruction in the very next issue slot.  LLVM's hazard recogniser inserts the
wait state only when a buffer store has NO register in its soffset field (GCNHazardRecognizer createsVALUHazard); on gfx950 the
race was lost with a register there too (the leaky relu's temporary landed on dword 2 of the previous record: 0.02-0.15 % of the
outputs wrong, differently on every run).

    isa_store_hazard.py file.s                  a `hipcc -S --cuda-device-only` listing
    isa_store_hazard.py libfisr_hip.so          the device code object is unbundled and disassembled (llvm-objcopy,
                                                clang-offload-bundler, llvm-objdump of /opt/rocm/lib/llvm/bin)
exit status 1 if anything is found."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(so_path):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def vregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(text, min_slots=1):
    """-> list of (function, store, slots between, overwriting instruction)"""
    func, body = None, []
    for l in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", l)
        if m:
            func = m.group(1)
            continue
        t = l.split("//")[0].strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        body.append((func, t))
    found = []
    for i, (fn, t) in enumerate(body):
        m = re.match(r"(buffer|global|scratch)_store_dwordx[34] ([^,]+), ([^,]+),", t)
        if not m:
            continue
        # buffer_store: vdata, vaddr, srsrc, soffset;  global / scratch_store: vaddr, vdata, saddr
        data = vregs(m.group(2) if m.group(1) == "buffer" else m.group(3))
        slots = 0
        for j in range(i + 1, min(i + 1 + 4, len(body))):
            if slots >= min_slots or body[j][0] != fn:
                break
            u = body[j][1]
            op = u.split()[0]
            if op.startswith("s_nop"):
                slots += int(u.split()[1]) + 1
                continue
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and len(u.split(None, 1)) > 1:
                if vregs(u.split(None, 1)[1].split(",")[0]) & data:
                    found.append((fn, t, slots, u))
                    break
            slots += 1
    return found


def main():
    path = sys.argv[1]
    text = disassemble(path) if path.endswith(".so") else open(path).read()
    found = scan(text)
    for fn, t, slots, u in found:
        print(f"{fn[:80]}: `{t[:70]}` then, {slots} slot(s) later, `{u[:60]}`")
    print(f"{found and len(found) or 0} store-data hazard(s): a vector instruction overwrites store data in the slot behind the store")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
