#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/gemm_dtl_192x256.inc (+ _clobbers.inc): the gfx950 assembly main loop of
gemm_bf16_kernel_dtl (gemm.hip, tile_cfg 75). The generated files are committed; the build does not run this script.

C[M][N] = A[M][K] . B[N][K]^T, bf16 in, f32 accumulate. Workgroup tile 192 x 256 (the DiT's 1536 rows x 8192 / 16384 columns are
exactly one / two rounds of 256 such tiles), four waves as 2 x 2 = ONE wave per SIMD, wave tile 96 x 128 = 6 x 8 accumulators of
v_mfma_f32_16x16x32_bf16 in 192 AGPRs, K-tiles of 64. Same LDS image as every other GEMM kernel of this library (128-byte rows,
16-byte chunks XOR-swizzled by (row >> 1) & 7 on the LDS-DMA source address, 8-row pieces of 1 KB per wave instruction), same
accumulator map as tools/gen_gemm_asm.py (a[(mi*8+ni)*4 ..]), so the epilogue of that kernel is reused.

Structure (what a 56 KB K-tile allows in 160 KB of LDS: two slots, no ring):
  * both k-steps' fragments of a K-tile live in REGISTERS (2 x (6 A + 8 B) x 4 VGPRs = 112): once a tile's 28 fragment reads have
    returned, its LDS slot is free, so the LDS-DMA of tile t+2 goes into the slot tile t is being multiplied from;
  * per K-tile t (slot p = t % 2), 96 MFMAs per wave, in one in-order stream:
      MFMA  1..14   k-step 0 products, one ds_read of a k-step-1 fragment (slot p) behind each
      after 26      s_waitcnt lgkmcnt(0) + s_barrier        every wave has tile t in registers: slot p is free   [WAR]
      MFMA 28..80   one LDS-DMA piece of tile t+2 -> slot p behind every 4th MFMA (14 pieces per wave: 6 of A, 8 of B)
      after 82      s_waitcnt vmcnt(14) + s_barrier         tile t+1 (staged one tile ago) has landed in slot p^1 [RAW]
      MFMA 83..96   one ds_read of a k-step-0 fragment of tile t+1 (slot p^1) behind each
    two barriers and no VALU instruction per K-tile; every wait is derived from the issue order, and check_wait_coverage()
    proves both hazards on the emitted text before the file is written.

Register map (per wave):
  v[0:23]  A fragments k-step 0   v[24:55]  B fragments k-step 0      v[56:79] / v[80:111]  the same for k-step 1
  v[112:115] slot-1 copies of the fragment addresses (fa0, fa1, fb0, fb1)        a[0:191] accumulators
  s[36:39] / s[40:43] A / B buffer descriptors   s46 tile counter   s[50:55] / s[56:63] scalar offsets of the A / B pieces
"""
import os
import re
import sys

MI, NI = 6, 8
NA, NB = 6, 8                 # LDS-DMA pieces per wave and K-tile
NL = NA + NB
A_BYTES = 192 * 128
STAGE = (192 + 256) * 128     # 57344
SET = (0, 56)                 # register sets of the two k-steps: A at +0 (6 x 4), B at +24 (8 x 4)
S1 = 112                      # slot-1 fragment addresses
NV = S1 + 4
SOFF = 50
READ_UNTIL = 14               # k-step-1 reads sit behind MFMAs 1..14
BAR1_AFTER = 26
DMA_FIRST, DMA_EVERY = 28, 4
BAR2_AFTER = 82
NEXT_FROM = 83


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def acc(mi, ni):
    b = (mi * NI + ni) * 4
    return f"a[{b}:{b + 3}]"


def frag_reg(ks, kind, i):
    return SET[ks] + (4 * i if kind == "A" else 24 + 4 * i)


# fragment read order of one k-step: what the MFMA stream (ni outer, mi inner) needs first comes first
READ_ORDER = [("B", 0)] + [("A", i) for i in range(MI)] + [("B", i) for i in range(1, NI)]


class Gen:
    def __init__(self):
        self.lines = []
        self.pending = []   # names of LDS reads in issue order (they return in order)

    def e(self, s):
        self.lines.append(s)

    def addr(self, slot, kind, ks):
        name = ("fa" if kind == "A" else "fb") + str(ks)
        if slot == 0:
            return f"%[{name}]"
        return f"v{S1 + (0 if kind == 'A' else 2) + ks}"

    def read(self, slot, ks, kind, i):
        self.e(f"ds_read_b128 {vr(frag_reg(ks, kind, i))}, {self.addr(slot, kind, ks)} offset:{i * 2048}")
        self.pending.append((ks, kind, i))

    def need(self, name):
        """counted wait: the read of fragment `name` (and every older one) has returned"""
        if name not in self.pending:
            return
        idx = len(self.pending) - 1 - self.pending[::-1].index(name)
        n = len(self.pending) - 1 - idx
        self.e(f"s_waitcnt lgkmcnt({min(n, 15)})")   # 4-bit field: a smaller count waits for more, never for less
        self.pending = self.pending[idx + 1:] if n <= 15 else self.pending[len(self.pending) - 15:]

    def dma(self, slot, i):
        if i < NA:
            dst = slot * STAGE + i * 4096
            return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[ao], s[36:39], s{SOFF + i} offen lds"]
        dst = slot * STAGE + A_BYTES + (i - NA) * 4096
        return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[bo], s[40:43], s{SOFF + i} offen lds"]

    def advance_k(self):
        for i in range(NL):
            self.e(f"s_add_u32 s{SOFF + i}, s{SOFF + i}, 128")

    def body(self, p):
        e = self.e
        e(f"; ================= K-tile body, slot {p}: tile t+2 -> slot {p}, entry reads of tile t+1 from slot {p ^ 1} =================")
        m = 0
        k1 = list(READ_ORDER)       # k-step-1 fragments of this tile, still to read
        nxt = list(READ_ORDER)      # k-step-0 fragments of the next tile
        dmas = [self.dma(p, i) for i in range(NL)]
        for ks in range(2):
            for ni in range(NI):
                for mi in range(MI):
                    m += 1
                    if mi == 0:
                        self.need((ks, "B", ni))
                    if ni == 0:
                        self.need((ks, "A", mi))
                    e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {vr(frag_reg(ks, 'A', mi))}, {vr(frag_reg(ks, 'B', ni))}, {acc(mi, ni)}")
                    if m <= READ_UNTIL and k1:
                        kind, i = k1.pop(0)
                        self.read(p, 1, kind, i)
                    if m == BAR1_AFTER:
                        assert not k1
                        e("s_waitcnt lgkmcnt(0)")          # this wave holds all of tile t in registers
                        self.pending = []
                        e("s_barrier")                      # ... and so does every other wave: slot p may be overwritten
                    if m >= DMA_FIRST and (m - DMA_FIRST) % DMA_EVERY == 0 and dmas:
                        for ins in dmas.pop(0):
                            e(ins)
                    if m == BAR2_AFTER:
                        assert not dmas
                        self.advance_k()
                        e(f"s_waitcnt vmcnt({NL})")         # all but the NL pieces just issued: tile t+1 has landed for this wave
                        e("s_barrier")                      # ... and for every other wave
                    if m >= NEXT_FROM and nxt:
                        kind, i = nxt.pop(0)
                        self.read(p ^ 1, 0, kind, i)
        assert m == 2 * MI * NI and not nxt

    def build(self):
        e = self.e
        for a, b in (("s36", "%[alo]"), ("s37", "%[ahi]"), ("s38", "%[arec]"), ("s39", "0x00020000"), ("s40", "%[blo]"),
                     ("s41", "%[bhi]"), ("s42", "%[brec]"), ("s43", "0x00020000"), ("s46", "%[nk]")):
            e(f"s_mov_b32 {a}, {b}")
        e(f"s_mov_b32 s{SOFF}, 0")
        for i in range(1, NA):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sa]")
        e(f"s_mov_b32 s{SOFF + NA}, 0")
        for i in range(NA + 1, NL):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sb]")
        for t in range(2):      # tiles 0, 1 -> slots 0, 1 (the weights are the HBM-cold operand: everything is in flight at once)
            for i in range(NL):
                for ins in self.dma(t, i):
                    e(ins)
            self.advance_k()
        for w, name in enumerate(("fa0", "fa1", "fb0", "fb1")):
            e(f"v_add_u32 v{S1 + w}, {STAGE}, %[{name}]")
        for i in range(MI * NI * 4):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        e(f"s_waitcnt vmcnt({NL})")     # tile 0 has landed
        e("s_barrier")
        for kind, i in READ_ORDER:
            self.read(0, 0, kind, i)
        entry = list(self.pending)
        e("10:")
        for p in range(2):
            self.pending = list(entry)
            self.body(p)
            assert self.pending == entry, "the loop body must leave the entry state it assumes"
            e("s_sub_u32 s46, s46, 1")
            e("s_cmp_eq_u32 s46, 0")
            e("s_cbranch_scc1 20f")
        e("s_branch 10b")
        e("20:")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")   # nothing may land in LDS after the epilogue scratch takes it over
        e("s_nop 7")
        e("s_nop 7")
        return self.lines


class WaitCoverageError(AssertionError):
    pass


def check_wait_coverage(lines, iterations=3):
    """Static proof of the LDS-DMA / ds_read ordering of the emitted stream (same rules as tools/gen_attn_w48.py):
      RAW  a ds_read of slot s needs every LDS-DMA issued into s to be retired by a counted vmcnt wait that is followed by a barrier;
      WAR  an LDS-DMA into slot s needs every earlier ds_read of s to be retired by an lgkmcnt wait that is followed by a barrier.
    Walks prologue + iterations x loop body + exit with branches not taken."""
    i10, ibr = lines.index("10:"), lines.index("s_branch 10b")
    seq = lines[:i10] + lines[i10 + 1:ibr] * iterations + lines[ibr + 1:]
    vm, lg, fills, reads = [], [], {0: [], 1: []}, {0: [], 1: []}
    m0 = None
    n_reads = n_dma = 0
    for pos, ins in enumerate(seq):
        mm = re.match(r"s_add_u32 m0, %\[wlds\], (\d+)", ins)
        if mm:
            m0 = int(mm.group(1))
            continue
        if ins.startswith("buffer_load_dwordx4") and ins.endswith("lds"):
            slot = m0 // STAGE
            for r in reads[slot]:
                if r["state"] != "fenced":
                    raise WaitCoverageError(f"WAR: LDS-DMA into slot {slot} at {pos} while the ds_read at {r['pos']} is only '{r['state']}'")
            reads[slot] = []
            op = {"state": "inflight", "pos": pos}
            vm.append(op)
            fills[slot].append(op)
            n_dma += 1
            m0 = None
            continue
        mm = re.match(r"ds_read_b128 v\[\d+:\d+\], (\S+) offset:(\d+)", ins)
        if mm:
            slot = 0 if mm.group(1).startswith("%[") else 1
            if not fills[slot]:
                raise WaitCoverageError(f"RAW: ds_read of slot {slot} at {pos} before anything was staged")
            for f in fills[slot]:
                if f["state"] != "visible":
                    raise WaitCoverageError(f"RAW: ds_read of slot {slot} at {pos} ({ins}) while the LDS-DMA at {f['pos']} is only '{f['state']}'")
            op = {"state": "issued", "pos": pos}
            lg.append(op)
            reads[slot].append(op)
            n_reads += 1
            continue
        if ins.startswith("s_waitcnt"):
            mv, ml = re.search(r"vmcnt\((\d+)\)", ins), re.search(r"lgkmcnt\((\d+)\)", ins)
            if mv:
                keep = int(mv.group(1))
                done, vm = (vm[:len(vm) - keep], vm[len(vm) - keep:]) if keep < len(vm) else ([], vm)
                for op in done:
                    op["state"] = "retired"
            if ml:
                keep = int(ml.group(1))
                done, lg = (lg[:len(lg) - keep], lg[len(lg) - keep:]) if keep < len(lg) else ([], lg)
                for op in done:
                    op["state"] = "done"
            continue
        if ins == "s_barrier":
            for s in (0, 1):
                for op in fills[s]:
                    if op["state"] == "retired":
                        op["state"] = "visible"
                for op in reads[s]:
                    if op["state"] == "done":
                        op["state"] = "fenced"
    if not n_reads or not n_dma:
        raise WaitCoverageError("checker saw no LDS traffic - the stream format changed")
    return {"reads": n_reads, "fills": n_dma, "instructions": len(seq)}


def check_operand_waits(lines):
    """Every MFMA operand fragment must have returned from LDS: walk one prologue + 3 loop bodies, tracking reads in issue order and
    the counted lgkmcnt waits; an MFMA that names a register whose read is still outstanding is an error."""
    i10, ibr = lines.index("10:"), lines.index("s_branch 10b")
    seq = lines[:i10] + lines[i10 + 1:ibr] * 3
    out = []  # (first register) of outstanding reads, issue order
    for pos, ins in enumerate(seq):
        mm = re.match(r"ds_read_b128 v\[(\d+):\d+\]", ins)
        if mm:
            out.append(int(mm.group(1)))
            continue
        ml = re.search(r"lgkmcnt\((\d+)\)", ins) if ins.startswith("s_waitcnt") else None
        if ml:
            keep = int(ml.group(1))
            out = out[len(out) - keep:] if keep < len(out) else out
            continue
        mm = re.match(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", ins)
        if mm:
            for r in (int(mm.group(1)), int(mm.group(2))):
                if r in out:
                    raise WaitCoverageError(f"MFMA at {pos} reads v[{r}:{r + 3}] while its ds_read is outstanding")
            continue
        # a read that overwrites a register an earlier-issued MFMA still has to read cannot happen here: MFMAs read their operands
        # at issue, and the stream is in order
    return True


def main():
    g = Gen()
    lines = g.build()
    if "--inject-raw-race" in sys.argv:   # checker self-test: the second barrier's wait leaves tile t+1 in flight
        k = lines.index(f"s_waitcnt vmcnt({NL})", lines.index("10:"))
        lines[k] = f"s_waitcnt vmcnt({2 * NL})"
    if "--inject-war-race" in sys.argv:   # checker self-test: the first barrier is dropped
        k = lines.index("s_barrier", lines.index("10:"))
        del lines[k]
    stats = check_wait_coverage(lines)
    check_operand_waits(lines)
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc")
    body = ["// GENERATED by tools/gen_gemm_asm_dtl.py - do not edit. gfx950 assembly main loop of gemm_bf16_kernel_dtl (gemm.hip).\n"]
    body += ['"' + ln + '\\n\\t"\n' for ln in lines]
    path = os.path.join(d, "gemm_dtl_192x256.inc")
    if "--check" in sys.argv:
        same = os.path.exists(path) and open(path).read() == "".join(body)
        print(f"wait coverage ok: {stats}; committed file {'matches' if same else 'DIFFERS'}")
        sys.exit(0 if same else 4)
    with open(path, "w") as f:
        f.write("".join(body))
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(MI * NI * 4)] + [f"s{i}" for i in range(36, 66)] + ["m0", "vcc", "scc", "memory"]
    with open(os.path.join(d, "gemm_dtl_192x256_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm_dtl.py - do not edit. Registers the assembly main loop assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    print(f"{len(lines)} lines, {sum(1 for ln in lines if 'v_mfma' in ln)} MFMAs; {stats}")


if __name__ == "__main__":
    main()
