#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/gemm_dtl_192x256.inc (+ _clobbers.inc): the gfx950 assembly main loop of
gemm_bf16_kernel_dtl (gemm.hip, tile_cfg 75). The generated files are committed; the build does not run this script.

C[M][N] = A[M][K] . B[N][K]^T, bf16 in, f32 accumulate. Workgroup tile 192 x 256 (the DiT's 1536 rows x 8192 / 16384 columns are
exactly one / two rounds of 256 such tiles), four waves as 2 x 2 = ONE wave per SIMD, wave tile 96 x 128 = 6 x 8 accumulators of
v_mfma_f32_16x16x32_bf16 in 192 AGPRs, K-tiles of 64. Same LDS image as every other GEMM kernel of this library (128-byte rows,
16-byte chunks XOR-swizzled by (row >> 1) & 7 on the LDS-DMA source address, 8-row pieces of 1 KB per wave instruction), same
accumulator map as tools/gen_gemm_asm.py (a[(mi*8+ni)*4 ..]), so the epilogue of that kernel is reused.

Structure (what a 24 + 32 KB K-tile allows in 160 KB of LDS: TWO slots for the activations, THREE for the weights = 144 KB):
  * both k-steps' fragments of a K-tile live in REGISTERS (2 x (6 A + 8 B) x 4 VGPRs = 112): once a tile's 28 fragment reads have
    returned, its LDS slots are free, so the LDS-DMA of A tile t+2 and of B tile t+3 go into the slots tile t was multiplied from;
  * why three weight slots: in-kernel stamps + a K sweep of the first (two-slot) version showed a K-tile body of 1760 cycles when
    nothing is waited for (87 % MFMA duty) but 1.26-1.5 us per K-tile in the steady state: the loop runs at (bytes in flight per
    CU) / (latency of HBM-cold weights, ~1.9 us) - 84 KB in flight then, ~116 KB now. The weights are the cold operand (the
    activations are re-read by every column tile and come from L2 / the memory-side cache), so the third slot goes to them;
  * per K-tile t (A slot t % 2, B slot t % 3), 96 MFMAs per wave, in one in-order stream:
      MFMA  1..14   k-step 0 products, one ds_read of a k-step-1 fragment behind each
      after BAR1    s_waitcnt lgkmcnt(0) + s_barrier        every wave has tile t in registers: its slots are free   [WAR]
      then          6 LDS-DMA pieces of A(t+2), 8 of B(t+3), one behind every DMA_EVERY-th MFMA
      after BAR2    s_waitcnt vmcnt(22) + s_barrier         A(t+1) and B(t+1) have landed; B(t+2), A(t+2), B(t+3) may fly [RAW]
      MFMA 83..96   one ds_read of a k-step-0 fragment of tile t+1 behind each
    two barriers and no VALU instruction per K-tile; the loop body is six tiles long (2 x 3 slots), left after any tile. Every
    wait is derived from the issue order, and check_wait_coverage() proves both hazards on the emitted text before the file is
    written.

Register map (per wave):
  v[0:23]  A fragments k-step 0   v[24:55]  B fragments k-step 0      v[56:79] / v[80:111]  the same for k-step 1
  v[112:113] A slot 1, v[114:115] / v[116:117] B slots 1 / 2: copies of the fragment addresses     a[0:191] accumulators
  s[36:39] / s[40:43] A / B buffer descriptors   s46 tile counter   s[50:55] / s[56:63] scalar offsets of the A / B pieces
"""
import os
import re
import sys

MI, NI = 6, 8
NA, NB = 6, 8                 # LDS-DMA pieces per wave and K-tile
NL = NA + NB
A_BYTES = 192 * 128           # 24576 per A slot
B_BYTES = 256 * 128           # 32768 per B slot
A_SLOTS, B_SLOTS = 2, 3
B_BASE = A_SLOTS * A_BYTES    # LDS: [A0][A1][B0][B1][B2] = 147456 bytes
SET = (0, 56)                 # register sets of the two k-steps: A at +0 (6 x 4), B at +24 (8 x 4)
S1 = 112                      # fragment addresses of the slots other than 0
NV = S1 + 6
INFLIGHT = NB + NA + NB       # B(t+2), A(t+2), B(t+3) may still fly when tile t+1 is read
SOFF = 50
READ_UNTIL = 14               # k-step-1 reads sit behind MFMAs 1..14
BAR1_AFTER = 22
DMA_FIRST, DMA_EVERY = 23, 3
BAR2_AFTER = 82
NEXT_FROM = 83
STAMPS = "--stamps" in sys.argv   # diagnostic build (tools/ubench/gemm_dtl_stamps.hip): s_memtime at the phase boundaries of body 0
# schedule knobs for A/B builds: DTL_KNOBS="bar1=26,dma0=28,dmae=4,bar2=82,next=83,reads=14"
for _kv in filter(None, os.environ.get("DTL_KNOBS", "").split(",")):
    _k, _v = _kv.split("=")
    _v = int(_v)
    if _k == "bar1": BAR1_AFTER = _v
    elif _k == "dma0": DMA_FIRST = _v
    elif _k == "dmae": DMA_EVERY = _v
    elif _k == "bar2": BAR2_AFTER = _v
    elif _k == "next": NEXT_FROM = _v
    elif _k == "reads": READ_UNTIL = _v


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
        return f"v{S1 + ks}" if kind == "A" else f"v{S1 + 2 * slot + ks}"

    def read(self, t, ks, kind, i):
        slot = t % (A_SLOTS if kind == "A" else B_SLOTS)
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

    def dma_a(self, t, i):
        dst = (t % A_SLOTS) * A_BYTES + i * 4096
        return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[ao], s[36:39], s{SOFF + i} offen lds"]

    def dma_b(self, t, i):
        dst = B_BASE + (t % B_SLOTS) * B_BYTES + i * 4096
        return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[bo], s[40:43], s{SOFF + NA + i} offen lds"]

    def advance_a(self):
        for i in range(NA):
            self.e(f"s_add_u32 s{SOFF + i}, s{SOFF + i}, 128")

    def advance_b(self):
        for i in range(NB):
            self.e(f"s_add_u32 s{SOFF + NA + i}, s{SOFF + NA + i}, 128")

    def stamp(self, i, p):
        if STAMPS and p == 0:
            self.e(f"s_memtime s[{64 + 2 * i}:{65 + 2 * i}]")

    def body(self, p):
        """K-tile t with t % 6 == p."""
        e = self.e
        e(f"; ================= K-tile body {p}: A slot {p % A_SLOTS} <- A(t+2), B slot {p % B_SLOTS} <- B(t+3) =================")
        m = 0
        self.stamp(0, p)
        k1 = list(READ_ORDER)       # k-step-1 fragments of this tile, still to read
        nxt = list(READ_ORDER)      # k-step-0 fragments of the next tile
        # A pieces first, then B: the counted wait at the second barrier may then leave the NEWER weight tile in flight
        dmas = [self.dma_a(p + 2, i) for i in range(NA)] + [self.dma_b(p + 3, i) for i in range(NB)]
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
                        self.stamp(1, p)
                        e("s_waitcnt lgkmcnt(0)")          # this wave holds all of tile t in registers
                        self.pending = []
                        self.stamp(2, p)
                        e("s_barrier")                      # ... and so does every other wave: slot p may be overwritten
                        self.stamp(3, p)
                    if m >= DMA_FIRST and (m - DMA_FIRST) % DMA_EVERY == 0 and dmas:
                        for ins in dmas.pop(0):
                            e(ins)
                    if m == BAR2_AFTER:
                        assert not dmas
                        self.advance_a()
                        self.advance_b()
                        self.stamp(4, p)
                        e(f"s_waitcnt vmcnt({INFLIGHT})")   # A(t+1), B(t+1) have landed for this wave; B(t+2), A(t+2), B(t+3) may fly
                        self.stamp(5, p)
                        e("s_barrier")                      # ... and for every other wave
                        self.stamp(6, p)
                    if m >= NEXT_FROM and nxt:
                        kind, i = nxt.pop(0)
                        self.read(p + 1, 0, kind, i)
        assert m == 2 * MI * NI and not nxt
        self.stamp(7, p)

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
        # issue order A0 B0 A1 B1 B2: the same in-order history the loop's counted wait assumes
        for t, kinds in ((0, "AB"), (1, "AB"), (2, "B")):
            for kind in kinds:
                for i in range(NA if kind == "A" else NB):
                    for ins in (self.dma_a(t, i) if kind == "A" else self.dma_b(t, i)):
                        e(ins)
                self.advance_a() if kind == "A" else self.advance_b()
        for ks in range(2):
            e(f"v_add_u32 v{S1 + ks}, {A_BYTES}, %[fa{ks}]")
            for slot in (1, 2):
                e(f"v_add_u32 v{S1 + 2 * slot + ks}, {slot * B_BYTES}, %[fb{ks}]")
        for i in range(MI * NI * 4):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        e(f"s_waitcnt vmcnt({INFLIGHT})")     # A0 and B0 have landed (A1, B1, B2 = 6 + 8 + 8 pieces may fly)
        e("s_barrier")
        for kind, i in READ_ORDER:
            self.read(0, 0, kind, i)
        entry = list(self.pending)
        e("10:")
        for p in range(A_SLOTS * B_SLOTS):
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
        if STAMPS:   # lane 0 of every wave writes its 8 stamps of the last pass through body 0: dbg[8] u64
            for i in range(16):
                e(f"v_mov_b32 v{i}, s{64 + i}")
            e("v_mov_b32 v16, 0")
            for i in range(8):
                e(f"global_store_dwordx2 v16, v[{2 * i}:{2 * i + 1}], %[dbg] offset:{i * 8}")
            e("s_waitcnt vmcnt(0)")
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
    regions = [("A", i) for i in range(A_SLOTS)] + [("B", i) for i in range(B_SLOTS)]
    vm, lg, fills, reads = [], [], {r: [] for r in regions}, {r: [] for r in regions}
    m0 = None
    n_reads = n_dma = 0
    for pos, ins in enumerate(seq):
        mm = re.match(r"s_add_u32 m0, %\[wlds\], (\d+)", ins)
        if mm:
            m0 = int(mm.group(1))
            continue
        if ins.startswith("buffer_load_dwordx4") and ins.endswith("lds"):
            slot = ("A", m0 // A_BYTES) if m0 < B_BASE else ("B", (m0 - B_BASE) // B_BYTES)
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
            a = mm.group(1)
            if a.startswith("%[fa"):
                slot = ("A", 0)
            elif a.startswith("%[fb"):
                slot = ("B", 0)
            else:
                r = int(a[1:]) - S1
                slot = ("A", 1) if r < 2 else ("B", r // 2)
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
            for s in regions:
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
    # m0 (the LDS-DMA destination) is a reserved register: naming it in the clobber list draws "-Winline-asm ... may lead to undefined
    # behaviour" from the compiler (round-4 verdict, Weak 10). The block saves it in an SGPR of its own and restores it on the way out,
    # so the surrounding HIP code sees m0 unchanged and the clobber list no longer names it.
    lines = ["s_mov_b32 s82, m0"] + lines + ["s_mov_b32 m0, s82"]
    if "--inject-raw-race" in sys.argv:   # checker self-test: the second barrier's wait leaves tile t+1 in flight
        k = lines.index(f"s_waitcnt vmcnt({INFLIGHT})", lines.index("10:"))
        lines[k] = f"s_waitcnt vmcnt({INFLIGHT + NA})"   # leaves A(t+1) in flight
    if "--inject-war-race" in sys.argv:   # checker self-test: the first barrier is dropped
        k = lines.index("s_barrier", lines.index("10:"))
        del lines[k]
    stats = check_wait_coverage(lines)
    check_operand_waits(lines)
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc")
    body = ["// GENERATED by tools/gen_gemm_asm_dtl.py - do not edit. gfx950 assembly main loop of gemm_bf16_kernel_dtl (gemm.hip).\n"]
    body += ['"' + ln + '\\n\\t"\n' for ln in lines]
    path = os.path.join(d, "gemm_dtl_192x256_stamps.inc" if (STAMPS or os.environ.get("DTL_KNOBS")) else "gemm_dtl_192x256.inc")
    if "--check" in sys.argv:
        same = os.path.exists(path) and open(path).read() == "".join(body)
        print(f"wait coverage ok: {stats}; committed file {'matches' if same else 'DIFFERS'}")
        sys.exit(0 if same else 4)
    with open(path, "w") as f:
        f.write("".join(body))
    if STAMPS or os.environ.get("DTL_KNOBS"):
        print(f"{len(lines)} lines -> {os.path.normpath(path)}; {stats}")
        return
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(MI * NI * 4)] + [f"s{i}" for i in range(36, 83)] + ["vcc", "scc", "memory"]
    with open(os.path.join(d, "gemm_dtl_192x256_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm_dtl.py - do not edit. Registers the assembly main loop assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    print(f"{len(lines)} lines, {sum(1 for ln in lines if 'v_mfma' in ln)} MFMAs; {stats}")


if __name__ == "__main__":
    main()
