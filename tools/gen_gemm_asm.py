#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/gemm_asm_192x256.inc (+ _clobbers.inc, _readout.inc): the gfx950 assembly main loop of
the one-wave-per-SIMD dense GEMM kernel gemm_bf16_kernel_asm (gemm.hip). The generated files are committed; the build does not
run this script.

C[M][N] = A[M][K] . B[N][K]^T, bf16 in, f32 accumulate. Workgroup tile 192 x 256, four waves as 2 x 2, wave tile 96 x 128 =
6 x 8 accumulators of v_mfma_f32_16x16x32_bf16 (192 AGPRs). K-tiles of 64. The LDS image of a K-tile is the ring kernel's
(128-byte rows, 16-byte pieces XOR-swizzled by (row>>1)&7 on the source address, 8-row pieces of 1 KB per wave instruction), so
the fragment addressing and the epilogue are shared with gemm_bf16_kernel_v2.

Why assembly: with one wave per SIMD the whole K-tile is ONE in-order instruction stream; the compiler-scheduled version of this
shape reached 600-680 TFLOP/s. Here the order is dictated. Global -> VGPR -> LDS staging (not LDS-DMA):
what hides the ~2 us weight-streaming latency is bytes in flight, and THREE 56 KB register sets + two LDS slots keep 2.5 K-tiles
in flight (an LDS-only ring of this tile holds one; the first version of this kernel with two register sets ran at 1.5 us per
K-tile = the load latency divided by its 1.5 tiles of lead).

  per K-tile t (LDS slot t%2; tile t+1 sits in register set (t+1)%3, tile t+2 is in flight to set (t+2)%3):
    k-step 0 (48 MFMAs): global loads of tile t+3 -> register set t%3 (one per 2 MFMAs; its old content, tile t, went to LDS during
                         tile t-1), the six A fragments of k-step 1, B fragments streamed two ahead
    k-step 1 (48 MFMAs): s_waitcnt vmcnt(28) (tile t+1 landed), its 14 ds_write_b128 into the other slot (one per 2 MFMAs),
                         after MFMA 36: lgkmcnt(0) + barrier, then the six A fragments of k-step 0 and the first two B fragments
                         of tile t+1 under the last 12 MFMAs
  The A fragments of a k-step are stationary (6 x 4 VGPRs, double-buffered over the k-steps), the B fragments stream through a
  ring of six, read five fragments (30 MFMAs, ~480 cycles) ahead: acc[mi][ni] += A[mi] . B[ni] with ni outer. With two fragments of
  lead the loop ran at 2430 cycles per K-tile against 1536 of MFMA time: every fragment waited for its LDS read, ~300 cycles under
  the load of four waves' reads and writes. 16 B fragments per tile shift the ring phase by four per tile (period 3) and the
  register sets rotate modulo 3, so the loop body is six tiles long; a scalar tile counter leaves it after any tile.

Register map (per wave):
  v[0:55] / v[56:111] / v[112:167]  register sets 0 / 1 / 2: one K-tile slice of this wave (6 A + 8 B pieces of 16 bytes per lane)
  v[168:191] / v[192:215]           A fragments of the even / odd k-step        v[216:239] B fragment ring (6)
  v[240:244]                        slot-1 copies of the LDS addresses (ds_write base, fa0, fa1, fb0, fb1)
  a[0:191]                          accumulators acc[mi][ni] = a[(mi*8+ni)*4 ...]
  s[36:39] / s[40:43] A / B buffer descriptors    s46 tile counter    s[50:55] / s[56:63] scalar offsets of the A / B pieces
"""
import os

import sys

BN = 128 if "--bn128" in sys.argv else 256   # workgroup tile 192 x BN; wave tile 96 x BN/2
MI, NI = 6, BN // 32
NB = BN // 32          # B pieces (8 rows x 128 B) per wave and K-tile
NL = 6 + NB            # global loads / ds_writes per wave and K-tile
SETSZ = 4 * NL
SET = (0, SETSZ, 2 * SETSZ)
FA = (3 * SETSZ, 3 * SETSZ + 24)
RING = 3 * SETSZ + 48
R = 6      # B fragment ring depth; a fragment is read R-1 fragments (6 MFMAs each) ahead of its first MFMA
S1 = RING + 4 * R  # wbase1, fa0_1, fa1_1, fb0_1, fb1_1
NV = S1 + 5
A_BYTES = 192 * 128
STAGE = (192 + BN) * 128
TAILF = 2 if NI == 8 else 1   # B fragments whose MFMAs run after the barrier (with the next tile's entry reads in between)
SOFF = 50  # s50..s55: A pieces, s56..: B pieces
NAME = f"192x{BN}"


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def acc(mi, ni):
    b = (mi * NI + ni) * 4
    return f"a[{b}:{b + 3}]"


STAMPS = "--stamps" in __import__("sys").argv


class Gen:
    def __init__(self):
        self.stamp_on = False
        self.lines = []
        self.lds_seq = 0   # LDS operations issued so far (reads and writes return in order)
        self.ready = {}    # fragment name -> sequence number of its read
        self.log = None    # when a list: names of the reads issued (to rebuild the entry state of the next tile body)

    def e(self, s):
        self.lines.append(s)

    def stamp(self, i):
        """--stamps build: s_memtime into s[64+2i : 65+2i] during tile body 0 (last pass survives), written out at the end."""
        if STAMPS and self.stamp_on:
            self.e(f"s_memtime s[{64 + 2 * i}:{65 + 2 * i}]")

    def lds(self, text, name=None):
        self.e(text)
        if name is not None:
            self.ready[name] = self.lds_seq
            if self.log is not None:
                self.log.append(name)
        self.lds_seq += 1

    def need(self, name):
        """s_waitcnt so that the read of fragment `name` has returned."""
        n = self.lds_seq - self.ready[name] - 1
        self.e(f"s_waitcnt lgkmcnt({min(n, 15)})")

    def wbase(self, slot):
        return "%[wb]" if slot == 0 else f"v{S1}"

    def fa(self, slot, ks):
        return f"%[fa{ks}]" if slot == 0 else f"v{S1 + 1 + ks}"

    def fb(self, slot, ks):
        return f"%[fb{ks}]" if slot == 0 else f"v{S1 + 3 + ks}"

    def read_a(self, slot, ks, mi):
        self.lds(f"ds_read_b128 {vr(FA[ks] + 4 * mi)}, {self.fa(slot, ks)} offset:{mi * 2048}", ("A", ks, mi))

    def read_b(self, slot, ks, ni, ring):
        self.lds(f"ds_read_b128 {vr(RING + 4 * ring)}, {self.fb(slot, ks)} offset:{ni * 2048}", ("B", ring))

    def load(self, set_, i):
        if i < 6:
            self.e(f"buffer_load_dwordx4 {vr(SET[set_] + 4 * i)}, %[ao], s[36:39], s{SOFF + i} offen")
        else:
            self.e(f"buffer_load_dwordx4 {vr(SET[set_] + 4 * i)}, %[bo], s[40:43], s{SOFF + i} offen")

    def advance_k(self):
        for i in range(NL):
            self.e(f"s_add_u32 s{SOFF + i}, s{SOFF + i}, 128")

    def write(self, set_, slot, i):
        off = i * 4096 if i < 6 else A_BYTES + (i - 6) * 4096
        self.lds(f"ds_write_b128 {self.wbase(slot)}, {vr(SET[set_] + 4 * i)} offset:{off}")

    ENTRY = [("B", 0), ("A", 0), ("B", 1), ("A", 1), ("B", 2), ("A", 2), ("B", 3), ("A", 3), ("A", 4), ("A", 5), ("B", 4)]

    def entry_read(self, slot, phase, k):
        """k-th of the reads a tile body expects to have been issued before it starts: B fragments 0..R-2 and the six A fragments
        of k-step 0, in the ENTRY order (B fragment R-2 last: its ring slot is the one fragment 14 of the previous tile uses)."""
        kind, x = self.ENTRY[k]
        if kind == "B":
            self.read_b(slot, x // NI, x % NI, (phase + x) % R)   # with four fragments per k-step, fragment 4 is k-step 1's first
        else:
            self.read_a(slot, 0, x)

    def tile(self, j):
        """Tile body j of the six-tile loop: LDS slot j%2, loads into set j%3, ds_writes from set (j+1)%3, ring phase j%3."""
        p, q = j % 2, 1 - j % 2
        ph = (2 * NI * j) % R  # ring slot of this tile's B fragment 0
        e = self.e
        e(f"; ================= K-tile body {j}: slot {p}, loads -> set {j % 3}, writes <- set {(j + 1) % 3} =================")
        loads = list(range(NL))
        writes = list(range(NL))
        na1 = 0
        pre = MI * (NI - TAILF)                     # MFMAs of k-step 1 before the barrier
        a1_at = [MI + (i * (MI * NI - MI)) // 6 + 1 for i in range(6)]   # MFMAs of k-step 0 after which an A(k-step 1) read goes
        wr_at = [1 + (i * (pre - 2)) // NL for i in range(NL)]          # MFMAs of k-step 1 after which a ds_write goes
        self.log = None
        tail_mfmas = []
        self.stamp_on = j == 0
        self.stamp(0)
        for ks in range(2):
            if ks == 1:
                self.stamp(1)
                e(f"s_waitcnt vmcnt({2 * NL})")  # tile t+1 has landed in its register set; tiles t+2 and t+3 may fly
                self.stamp(2)
            for ni in range(NI):
                f = ks * NI + ni  # B fragment index within the tile
                ring = (ph + f) % R
                tail = ks == 1 and ni >= NI - TAILF  # after the barrier
                for mi in range(MI):
                    m = ni * MI + mi  # MFMA index within the k-step
                    text = f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {vr(FA[ks] + 4 * mi)}, {vr(RING + 4 * ring)}, {acc(mi, ni)}"
                    if tail:
                        tail_mfmas.append(text)
                        continue
                    if mi == 0:
                        self.need(("B", ring))
                    if ni == 0:
                        self.need(("A", ks, mi))
                    e(text)
                    if mi == 0:
                        nf = f + R - 1  # B fragment R-1 ahead, into the ring slot of fragment f-1 (all its MFMAs are issued)
                        if nf < 2 * NI:
                            self.read_b(p, nf // NI, nf % NI, (ph + nf) % R)
                    if ks == 0:
                        if m % 2 == 1 and loads:
                            self.load(j % 3, loads.pop(0))
                        while na1 < MI and a1_at[na1] <= m:
                            self.read_a(p, 1, na1)
                            na1 += 1
                    else:
                        while writes and wr_at[NL - len(writes)] <= m:
                            self.write((j + 1) % 3, q, writes.pop(0))
                if ks == 0 and ni == NI - 1:
                    assert na1 == MI and not loads, (na1, loads)
                if ks == 1 and ni == NI - TAILF - 1:
                    assert not writes, writes
                    self.stamp(3)
                    e("s_waitcnt lgkmcnt(0)")
                    e("s_barrier")
                    self.stamp(4)
        self.advance_k()
        # tail: the last 12 MFMAs (B fragments 14, 15 are in their ring slots) with the entry reads of tile t+1 between them. The
        # next tile's fragment 0 goes to the ring slot fragment 13 used (free), its fragment 1 to the slot of fragment 14 - read
        # only once fragment 14's six MFMAs are issued.
        nph = (ph + 2 * NI) % R
        self.log = []
        k = 0
        ne = len(self.ENTRY)
        for idx, text in enumerate(tail_mfmas):
            e(text)
            # one entry read per MFMA; the last one (B fragment R-2, ring slot of fragment 14) only after fragment 14's six MFMAs
            # the last entry read (B fragment R-2) reuses the ring slot of fragment 2NI-2: with two tail fragments it waits for that
            # fragment's six MFMAs; entry reads go two per MFMA when the tail is one fragment long
            per = 1 if TAILF == 2 else 2
            while k < ne and (k < per * (idx + 1) if (k < ne - 1 or TAILF == 1) else idx >= MI and k < per * (idx + 1)):
                self.entry_read(q, nph, k)
                k += 1
        while k < ne:
            self.entry_read(q, nph, k)
            k += 1
        self.stamp(5)
        if STAMPS and self.stamp_on:
            e("s_waitcnt lgkmcnt(0)")  # the stamps themselves (scalar memory) before the next body's counted waits
            self.lds_seq = 0
            self.log_reset = True

    def build(self):
        e = self.e
        e("s_mov_b32 s36, %[alo]")
        e("s_mov_b32 s37, %[ahi]")
        e("s_mov_b32 s38, %[arec]")
        e("s_mov_b32 s39, 0x00020000")
        e("s_mov_b32 s40, %[blo]")
        e("s_mov_b32 s41, %[bhi]")
        e("s_mov_b32 s42, %[brec]")
        e("s_mov_b32 s43, 0x00020000")
        e("s_mov_b32 s46, %[nk]")
        e(f"s_mov_b32 s{SOFF}, 0")
        for i in range(1, 6):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sa]")   # piece i of A: rows 32 i .. of the tile
        e(f"s_mov_b32 s{SOFF + 6}, 0")
        for i in range(7, NL):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sb]")
        e(f"v_add_u32 v{S1}, {STAGE}, %[wb]")
        for ks in range(2):
            e(f"v_add_u32 v{S1 + 1 + ks}, {STAGE}, %[fa{ks}]")
            e(f"v_add_u32 v{S1 + 3 + ks}, {STAGE}, %[fb{ks}]")
        for i in range(MI * NI * 4):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        for t in range(3):  # tiles 0, 1, 2 -> register sets 0, 1, 2
            for i in range(NL):
                self.load(t, i)
            self.advance_k()
        e(f"s_waitcnt vmcnt({2 * NL})")
        for i in range(NL):
            self.write(0, 0, i)
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        self.log = []
        for k in range(len(self.ENTRY)):
            self.entry_read(0, 0, k)
        entry = list(self.log)
        e("10:")
        for j in range(6):
            # entry state: the last LDS operations issued were the entry reads of this tile, in `entry` order
            self.lds_seq = len(entry)
            self.ready = {name: k for k, name in enumerate(entry)}
            self.tile(j)
            entry = list(self.log)
            e("s_sub_u32 s46, s46, 1")
            e("s_cmp_eq_u32 s46, 0")
            e("s_cbranch_scc1 20f")
        e("s_branch 10b")
        e("20:")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_nop 7")
        e("s_nop 7")
        if STAMPS:
            for i in range(12):
                e(f"v_mov_b32 v{i}, s{64 + i}")
            e("v_mov_b32 v12, 0")
            for i in range(6):
                e(f"global_store_dwordx2 v12, v[{2 * i}:{2 * i + 1}], %[dbg] offset:{i * 8}")
            e("s_waitcnt vmcnt(0)")
        return self.lines


def main():
    g = Gen()
    lines = g.build()
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc")
    with open(os.path.join(d, f"gemm_asm_{NAME}_stamps.inc" if STAMPS else f"gemm_asm_{NAME}.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py - do not edit. gfx950 assembly main loop of gemm_bf16_kernel_asm (gemm.hip).\n")
        for ln in lines:
            f.write('"' + ln + '\\n\\t"\n')
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(MI * NI * 4)] + [f"s{i}" for i in range(36, 76)] + ["vcc", "scc", "memory"]
    with open(os.path.join(d, f"gemm_asm_{NAME}_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py - do not edit. Registers the assembly main loop assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    # Accumulator hand-over to the C++ epilogue: three assembly blocks, each writes two 16-row slabs (32 rows x 128 columns f32 =
    # 16 KB per wave) of a[0:191] into the wave's LDS scratch in row-major order; C++ then walks the rows in a ROLLED loop. (Reading
    # the accumulators into C++ values made the compiler unroll the generic epilogue over 6 slabs x 8 row pairs: ~2000 basic
    # blocks of straight-line code run once by one wave per SIMD - 53 000 cycles of instruction fetch per tile.)
    # lane (c = lane & 15, g = lane >> 4) holds acc[mi][ni][r] = C[mi*16 + 4g + r][ni*16 + c]; %[sb] = scratch + ((4g)*128 + c)*4.
    with open(os.path.join(d, f"gemm_asm_{NAME}_dump.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm.py - do not edit. if constexpr (grp == G): slabs 2G, 2G+1 of a[0:191] -> LDS scratch.\n")
        for grp in range(3):
            ins = []
            for half in range(2):
                mi = 2 * grp + half
                for ni in range(NI):
                    base = (mi * NI + ni) * 4
                    for r in range(4):
                        ins.append(f"v_accvgpr_read_b32 v{r}, a{base + r}")
                    ins.append("s_nop 0")
                    for r in range(4):
                        ins.append(f"ds_write_b32 %[sb], v{r} offset:{((half * 16 + r) * (16 * NI) + ni * 16) * 4}")
            ins.append("s_waitcnt lgkmcnt(0)")
            body = "".join(x + "\\n\\t" for x in ins)
            f.write(f'if constexpr (grp == {grp}) asm volatile("{body}" : : [sb] "v"(scr_lane) : "v0", "v1", "v2", "v3", "memory");\n')
    n_mfma = sum(1 for ln in lines if "v_mfma" in ln)
    print(f"{len(lines)} lines, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
