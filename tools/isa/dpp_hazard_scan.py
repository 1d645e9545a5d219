"""DPP read-after-VALU-write hazard scan of one kernel of an llvm-objdump listing:

    python tools/isa/dpp_hazard_scan.py /tmp/r1d.s r1d_kernelILi64ELi4

gfx9 (gfx950 included) needs 2 wait states between a VALU instruction that writes a VGPR and a DPP instruction that reads
it as its DPP operand (src0).  The compiler's hazard recogniser inserts them for code it schedules itself; it does not look
INSIDE an asm statement, and csrc/quad_narrow.h / resnet1d.hip carry hand-written v_*_dpp blocks (GLDM_DPP8, fmac_ror,
dpp_max) whose correctness rests on an s_nop in front and on register allocation never placing a v_mov copy or a reload
right before them.  This scan walks the final ISA: every instruction is one wait state, `s_nop N` is N + 1; a DPP
instruction whose src0 register was written by a VALU instruction fewer than 2 wait states earlier is reported.  Exit code 1
on any hit (tools/isa/lint.sh fails then).

Second check (round 6): a matrix instruction's result may be read by a VALU instruction only some wait states later.  The
compiler keeps that distance for its own instructions -- the smallest it leaves in this library's listings is 6 behind a
v_mfma_f32_16x16x4_f32 and 8 behind a v_mfma_f32_16x16x32_f16 (other instructions in between, each counted as one) -- so
anything closer than kMfmaWait can only be an asm statement that takes accumulators as operands straight out of the matrix
pipe (csrc/quad_narrow.h: pos_max8 records the one this found, 3 wait states).  A register overwritten in between no longer
counts."""
kMfmaWait = {"v_mfma_f32_16x16x4_f32": 6}
kMfmaWaitDefault = 7
import re
import sys


def regs(tok):
    """'v12' -> {12}; 'v[4:7]' -> {4,5,6,7}; anything else -> set()."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main(path, kernel):
    inside, hits, n_dpp = False, [], 0
    recent = []   # (wait states since, written vgprs, text) of the last VALU writes
    mrecent, mhits = [], []   # the same for matrix instructions
    for ln, line in enumerate(open(path), 1):
        if re.match(r"^[0-9a-f]+ <.*>:$", line.strip()):
            inside = kernel in line
            recent = []
            continue
        if not inside:
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        parts = text.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        if op == "s_nop":
            k = int(ops[0], 0) + 1 if ops else 1
            recent = [(w + k, r, t) for w, r, t in recent]
            mrecent = [(w + k, r, t) for w, r, t in mrecent if w + k < 16]
            continue
        is_dpp = op.endswith("_dpp") or " row_" in text or "quad_perm" in text or "row_bcast" in text or "wave_" in text
        if is_dpp and len(ops) >= 2:
            n_dpp += 1
            src0 = regs(ops[1].split()[0])
            for w, r, t in recent:
                if w < 2 and (src0 & r):
                    hits.append((ln, text, t, w))
        is_mfma = op.startswith(("v_mfma", "v_smfmac"))
        if op.startswith("v_") and not is_mfma and len(ops) >= 2:   # VALU read of a fresh matrix result (a matrix instruction's own srcC is interlocked)
            srcs = set()
            for o in ops[1:]:
                srcs |= regs(o.split()[0].strip("|-"))
            for w, r, t in mrecent:
                if (srcs & r) and w < kMfmaWait.get(t.split()[0], kMfmaWaitDefault):
                    mhits.append((ln, text, t, w))
        # a register overwritten by anything else no longer holds the matrix result
        dst = set()
        if not is_mfma and ops and op.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load", "flat_load")) and \
                not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dst = regs(ops[0].split()[0])
        mrecent = [(w + 1, r - dst, t) for w, r, t in mrecent if w + 1 < 16 and (r - dst)]
        if is_mfma and ops:
            mrecent.append((0, regs(ops[0].split()[0]), text))
        # every instruction is one wait state for what came before it
        recent = [(w + 1, r, t) for w, r, t in recent if w + 1 < 3]
        if op.startswith("v_") and ops and not op.startswith(("v_readlane", "v_readfirstlane", "v_cmp", "v_mfma", "v_smfmac")):
            dst = regs(ops[0].split()[0])
            if dst:
                recent.append((0, dst, text))
    print(f"{kernel}: {n_dpp} DPP instructions, {len(hits)} VALU-write -> DPP-read pairs closer than 2 wait states")
    for ln, text, t, w in hits[:20]:
        print(f"  line {ln}: `{text}` reads a register written {w} wait state(s) earlier by `{t}`")
    print(f"{kernel}: {len(mhits)} VALU reads of a matrix instruction's result closer than the compiler ever leaves them")
    for ln, text, t, w in mhits[:20]:
        print(f"  line {ln}: `{text}` reads a register written {w} wait state(s) earlier by `{t}`")
    return 1 if hits or mhits else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
