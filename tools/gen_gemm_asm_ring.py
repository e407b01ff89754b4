#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/gemm_asm_ring_192x128.inc (+ _clobbers.inc): the gfx950 assembly main loop of
gemm_bf16_kernel_asm_ring (gemm.hip, tile_cfg 73). The generated files are committed; the build does not run this script.

Same wave tile as tools/gen_gemm_asm.py --bn128 (workgroup tile 192 x 128, four waves as 2 x 2, 96 x 64 per wave = 6 x 4
accumulators of v_mfma_f32_16x16x32_bf16 in 96 AGPRs, A fragments of a k-step stationary, B fragments through a ring), but the
K-tiles are staged the way the 8-wave ring kernel does it: LDS-DMA (buffer_load ... lds) into a ring of FOUR 40 KB slots, three
K-tiles ahead, with a counted vmcnt. Why: the register-staged version pays ~29 cycles of issue per KB staged (buffer load + ds_write)
against ~5 for an LDS-DMA piece placed between MFMAs (measured in the attention kernel), and the 8-wave ring kernel is bound by LDS
traffic (597 LDS bytes per MFMA, 100 % of its ideal MFMA time) where this wave tile needs 427.

  per K-tile t (LDS slot t%4):
    k-step 0 (24 MFMAs): LDS-DMA of tile t+3 -> slot (t+3)%4 (10 pieces per wave, one per 2 MFMAs; the slot held tile t-1, which
                         nobody reads after the previous tile's barrier), the six A fragments of k-step 1, B fragments streamed
    k-step 1 (24 MFMAs): after MFMA 18: s_waitcnt vmcnt(20) (tile t+1 landed; t+2, t+3 may fly) + lgkmcnt(0) + barrier, then the entry
                         fragment reads of tile t+1 under the last 6 MFMAs
  8 B fragments per tile and a ring of 4: no phase shift; the loop body is four tiles long (the slots), left after any tile.

--bdirect (tile_cfg 74, gemm_asm_hybrid_192x128.inc): A through the same ring (four 24 KB slots), B never in LDS: fragment f of K-tile
t+4 is loaded into the registers of fragment f of tile t right after the six MFMAs that read them (v[48:175], four sets of eight
fragments). LDS-DMA pieces and fragment loads share one in-order vmcnt, so every wait is derived from the issue order (Gen.vm):
eight tiles are generated and thrown away to obtain the steady-state history, then the four real ones. Bit-exact, half the speed of
the ring variant on row-major weights.

Register map (per wave):
  v[0:23] / v[24:47]   A fragments of the even / odd k-step      v[48:63] B fragment ring (4)
  v[64:75]             fragment addresses of slots 1..3 (fa0, fa1, fb0, fb1 each)
  a[0:95]              accumulators acc[mi][ni] = a[(mi*4+ni)*4 ...]
  s[36:39] / s[40:43] A / B buffer descriptors    s46 tile counter    s[50:55] / s[56:59] scalar offsets of the A / B pieces
"""
import os
import sys

MI, NI = 6, 4
FA = (0, 24)
RING = 48
R = 4
BDIRECT = "--bdirect" in sys.argv   # tile_cfg 74: A through the LDS-DMA ring, B as fragment-layout loads straight into registers
A_BYTES = 192 * 128
if BDIRECT:
    BSET = 48           # v48..v175: four sets (K-tiles t .. t+3) of eight B fragments
    SADDR = 176         # v176.. : fa0, fa1 for slots 1, 2, 3
    NV = SADDR + 6
    STAGE = A_BYTES     # 24576: the ring holds A only
    NL = 6              # LDS-DMA pieces per wave and K-tile (A)
else:
    SADDR = 64          # v64.. : fa0, fa1, fb0, fb1 for slots 1, 2, 3
    NV = SADDR + 12
    STAGE = (192 + 128) * 128   # 40960
    NL = 10             # LDS-DMA pieces per wave and K-tile (6 A + 4 B)
SOFF = 50
STAMPS = "--stamps" in sys.argv
ABLATE = set(filter(None, os.environ.get("GEMM_ABLATE", "").split(",")))   # nodma, nolds, nobar: timing experiments only (wrong results)


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def acc(mi, ni):
    b = (mi * NI + ni) * 4
    return f"a[{b}:{b + 3}]"


class Gen:
    ENTRY = ([("A", i) for i in range(6)] if BDIRECT else
             [("B", 0), ("A", 0), ("A", 1), ("B", 1), ("A", 2), ("A", 3), ("B", 2), ("A", 4), ("A", 5)])

    def __init__(self):
        self.lines = []
        self.lds_seq = 0
        self.ready = {}
        self.log = None
        self.in_loop = False
        self.vm = []        # vector-memory operations in issue order (LDS-DMA pieces and B fragment loads count in one in-order counter)
        self.dry = False    # dry run of the tiles in front of the loop: fills self.vm with the steady-state history, emits nothing

    def vm_issue(self, name):
        self.vm.append(name)

    def vm_count(self, name):
        """vmcnt that guarantees `name` (its last piece) has landed: the number of operations issued after it."""
        hits = [i for i, nm in enumerate(self.vm) if nm == name]
        if not hits:
            assert self.dry, name   # only the thrown-away tiles may ask for operations older than the history
            return 0
        n = len(self.vm) - 1 - max(hits)
        assert n <= 63, n
        return n

    def e(self, s):
        if self.dry:
            return
        if self.in_loop and (("nodma" in ABLATE and s.startswith("buffer_load")) or ("nobar" in ABLATE and s == "s_barrier")):
            return
        if self.in_loop and "nolds" in ABLATE and s.startswith("s_waitcnt lgkmcnt"):
            return
        self.lines.append(s)

    def lds(self, text, name):
        if "nolds" in ABLATE and self.in_loop:
            return
        self.e(text)
        self.ready[name] = self.lds_seq
        if self.log is not None:
            self.log.append(name)
        self.lds_seq += 1

    def stamp(self, i, j):
        if STAMPS and j == 0:
            self.e(f"s_memtime s[{64 + 2 * i}:{65 + 2 * i}]")

    def need(self, name):
        if "nolds" in ABLATE and self.in_loop:
            return
        n = self.lds_seq - self.ready[name] - 1
        self.e(f"s_waitcnt lgkmcnt({min(n, 15)})")

    def addr(self, slot, which):  # which: 0 fa0, 1 fa1, 2 fb0, 3 fb1
        if BDIRECT:
            return ("%[fa0]", "%[fa1]")[which] if slot == 0 else f"v{SADDR + (slot - 1) * 2 + which}"
        return ("%[fa0]", "%[fa1]", "%[fb0]", "%[fb1]")[which] if slot == 0 else f"v{SADDR + (slot - 1) * 4 + which}"

    def bfrag(self, t, f):
        return vr(BSET + (t % 4) * 32 + f * 4)

    def load_b(self, t, f):
        """B fragment f = (k-step, 16-column block) of K-tile t: lane l gets B[16 ni + (l & 15)][32 ks + 8 (l >> 4) ..+7], 16 bytes."""
        ks, ni = f // NI, f % NI
        if "packedb" in ABLATE:   # timing experiment: what fragment-ordered weights (1 KB contiguous per fragment) would cost
            self.e(f"buffer_load_dwordx4 {self.bfrag(t, f)}, %[bfo], s[40:43], s{SOFF + 6 + ni} offen offset:{ks * 1024}")
            self.vm_issue(("B", t, f))
            return
        self.e(f"buffer_load_dwordx4 {self.bfrag(t, f)}, %[bfo], s[40:43], s{SOFF + 6 + ni} offen offset:{ks * 64}")
        self.vm_issue(("B", t, f))

    def read_a(self, slot, ks, mi):
        self.lds(f"ds_read_b128 {vr(FA[ks] + 4 * mi)}, {self.addr(slot, ks)} offset:{mi * 2048}", ("A", ks, mi))

    def read_b(self, slot, ks, ni, ring):
        self.lds(f"ds_read_b128 {vr(RING + 4 * ring)}, {self.addr(slot, 2 + ks)} offset:{ni * 2048}", ("B", ring))

    def dma(self, slot, i):
        """Piece i of this wave (A: 0..5, B: 6..9) of the tile at the current scalar offsets into ring slot `slot`."""
        if i < 6:
            dst = slot * STAGE + i * 4096
            return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[ao], s[36:39], s{SOFF + i} offen lds"]
        dst = slot * STAGE + A_BYTES + (i - 6) * 4096
        return [f"s_add_u32 m0, %[wlds], {dst}", "s_nop 0", f"buffer_load_dwordx4 %[bo], s[40:43], s{SOFF + i} offen lds"]

    def advance_k(self):
        for i in range(NL):
            self.e(f"s_add_u32 s{SOFF + i}, s{SOFF + i}, 128")

    def advance_b(self):
        for i in range(NI):
            self.e(f"s_add_u32 s{SOFF + 6 + i}, s{SOFF + 6 + i}, {16384 if 'packedb' in ABLATE else 128}")

    def entry_read(self, slot, k):
        kind, x = self.ENTRY[k]
        if kind == "B":
            self.read_b(slot, x // NI, x % NI, x % R)
        else:
            self.read_a(slot, 0, x)

    def tile(self, j):
        p, q = j % 4, (j + 1) % 4
        e = self.e
        e(f"; ================= K-tile body {j}: slot {p}, LDS-DMA -> slot {(j + 3) % 4} =================")
        dmas = [self.dma((j + 3) % 4, i) for i in range(NL)]
        a1_at = [MI + (i * (MI * NI - MI)) // 6 + 1 for i in range(6)]
        na1 = 0
        self.log = None
        tail = []
        self.stamp(0, j)
        for ks in range(2):
            if ks == 1:
                self.stamp(1, j)
            for ni in range(NI):
                f = ks * NI + ni
                ring = f % R
                is_tail = ks == 1 and ni == NI - 1
                for mi in range(MI):
                    m = ni * MI + mi
                    text = f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {vr(FA[ks] + 4 * mi)}, {vr(RING + 4 * ring)}, {acc(mi, ni)}"
                    if is_tail:
                        tail.append(text)
                        continue
                    if mi == 0:
                        self.need(("B", ring))
                    if ni == 0:
                        self.need(("A", ks, mi))
                    e(text)
                    if mi == 0:
                        nf = f + R - 1
                        if nf < 2 * NI:
                            self.read_b(p, nf // NI, nf % NI, nf % R)
                    if ks == 0:
                        if m % 2 == 1 and dmas:
                            for ins in dmas.pop(0):
                                e(ins)
                        while na1 < MI and a1_at[na1] <= m:
                            self.read_a(p, 1, na1)
                            na1 += 1
                if ks == 0 and ni == NI - 1:
                    assert na1 == MI and not dmas, (na1, len(dmas))
                if ks == 1 and ni == NI - 2:
                    self.stamp(2, j)
                    if STAMPS:
                        e(f"s_waitcnt vmcnt({2 * NL})")
                        self.stamp(3, j)
                        e("s_waitcnt lgkmcnt(0)")
                    else:
                        e(f"s_waitcnt vmcnt({2 * NL}) lgkmcnt(0)")   # tile t+1 has landed for this wave; tiles t+2, t+3 may fly
                    e("s_barrier")
                    self.stamp(4, j)
        self.advance_k()
        self.log = []
        k = 0
        ne = len(self.ENTRY)
        for idx, text in enumerate(tail):
            e(text)
            while k < ne and k < 2 * (idx + 1):
                self.entry_read(q, k)
                k += 1
        while k < ne:
            self.entry_read(q, k)
            k += 1
        self.stamp(5, j)
        if STAMPS and j == 0:
            e("s_waitcnt lgkmcnt(0)")

    # ------------------------------------------------------------------ tile_cfg 74: A through the ring, B straight to registers
    def tile_bdirect(self, T):
        """K-tile T (LDS slot T % 4, B register set T % 4). Issues the LDS-DMA of A tile T+3 and, right after the six MFMAs that
        consumed fragment f of this tile, the load of fragment f of tile T+4 into the same registers. Every wait is a counted vmcnt:
        the number of vector-memory operations issued after the one waited for (self.vm holds the issue order)."""
        p, q = T % 4, (T + 1) % 4
        e = self.e
        e(f"; ================= K-tile body {T % 4}: A slot {p}, LDS-DMA -> slot {(T + 3) % 4}, B set {T % 4} =================")
        dmas = [self.dma((T + 3) % 4, i) for i in range(NL)]
        a1_at = [MI + (i * (MI * NI - MI)) // 6 + 1 for i in range(6)]
        na1 = 0
        self.log = None
        tail = []
        for ks in range(2):
            for ni in range(NI):
                f = ks * NI + ni
                is_tail = ks == 1 and ni == NI - 1
                for mi in range(MI):
                    m = ni * MI + mi
                    text = f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {vr(FA[ks] + 4 * mi)}, {self.bfrag(T, f)}, {acc(mi, ni)}"
                    if is_tail:
                        tail.append(text)
                        continue
                    if mi == 0:
                        e(f"s_waitcnt vmcnt({self.vm_count(('B', T, f))})")
                    if ni == 0:
                        self.need(("A", ks, mi))
                    e(text)
                    if ks == 0:
                        if m % 4 == 1 and dmas:
                            for ins in dmas.pop(0):
                                e(ins)
                            self.vm_issue(("A", T + 3))
                        while na1 < MI and a1_at[na1] <= m:
                            self.read_a(p, 1, na1)
                            na1 += 1
                if not is_tail:
                    self.load_b(T + 4, f)
                if ks == 0 and ni == NI - 1:
                    assert na1 == MI and not dmas, (na1, len(dmas))
                if ks == 1 and ni == NI - 2:
                    e(f"s_waitcnt vmcnt({self.vm_count(('A', T + 1))}) lgkmcnt(0)")   # A tile t+1 has landed for this wave
                    e("s_barrier")
        self.advance_k()
        self.log = []
        k = 0
        ne = len(self.ENTRY)
        e(f"s_waitcnt vmcnt({self.vm_count(('B', T, 2 * NI - 1))})")
        for idx, text in enumerate(tail):
            e(text)
            while k < ne and k < idx + 1:
                self.entry_read(q, k)
                k += 1
        while k < ne:
            self.entry_read(q, k)
            k += 1
        self.load_b(T + 4, 2 * NI - 1)
        self.advance_b()

    def build_bdirect(self):
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
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sa]")
        e(f"s_mov_b32 s{SOFF + 6}, 0")
        for i in range(7, 10):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sb16]")
        for t in range(4):  # B tiles 0..3 -> register sets 0..3 (the weights first: they are the HBM-cold operand)
            for f in range(2 * NI):
                self.load_b(t, f)
            self.advance_b()
        for t in range(3):  # A tiles 0, 1, 2 -> slots 0, 1, 2
            for i in range(NL):
                for ins in self.dma(t, i):
                    e(ins)
            self.advance_k()
        for slot in range(1, 4):
            for w, name in enumerate(("fa0", "fa1")):
                e(f"v_add_u32 v{SADDR + (slot - 1) * 2 + w}, {slot * STAGE}, %[{name}]")
        for i in range(MI * NI * 4):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        self.log = []
        for k in range(len(self.ENTRY)):
            self.entry_read(0, k)
        entry = list(self.log)
        # steady-state history of the vector-memory counter: eight tiles generated and thrown away
        self.vm = []
        lines = self.lines
        self.dry = True
        for T in range(-8, 0):
            self.lds_seq = len(entry)
            self.ready = {name: k for k, name in enumerate(entry)}
            self.tile_bdirect(T)
            entry = list(self.log)
        self.dry = False
        assert self.lines is lines
        e("10:")
        self.in_loop = True
        for T in range(4):
            self.lds_seq = len(entry)
            self.ready = {name: k for k, name in enumerate(entry)}
            self.tile_bdirect(T)
            entry = list(self.log)
            e("s_sub_u32 s46, s46, 1")
            e("s_cmp_eq_u32 s46, 0")
            e("s_cbranch_scc1 20f")
        self.in_loop = False
        e("s_branch 10b")
        e("20:")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_nop 7")
        e("s_nop 7")
        return self.lines

    def build(self):
        if BDIRECT:
            return self.build_bdirect()
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
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sa]")
        e(f"s_mov_b32 s{SOFF + 6}, 0")
        for i in range(7, NL):
            e(f"s_add_u32 s{SOFF + i}, s{SOFF + i - 1}, %[sb]")
        for t in range(3):  # tiles 0, 1, 2 -> slots 0, 1, 2
            for i in range(NL):
                for ins in self.dma(t, i):
                    e(ins)
            self.advance_k()
        for slot in range(1, 4):
            for w, name in enumerate(("fa0", "fa1", "fb0", "fb1")):
                e(f"v_add_u32 v{SADDR + (slot - 1) * 4 + w}, {slot * STAGE}, %[{name}]")
        for i in range(MI * NI * 4):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        e(f"s_waitcnt vmcnt({2 * NL})")
        e("s_barrier")
        self.log = []
        for k in range(len(self.ENTRY)):
            self.entry_read(0, k)
        entry = list(self.log)
        e("10:")
        self.in_loop = True
        for j in range(4):
            self.lds_seq = len(entry)
            self.ready = {name: k for k, name in enumerate(entry)}
            self.tile(j)
            entry = list(self.log)
            e("s_sub_u32 s46, s46, 1")
            e("s_cmp_eq_u32 s46, 0")
            e("s_cbranch_scc1 20f")
        self.in_loop = False
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
    lines = ["s_mov_b32 s76, m0"] + g.build() + ["s_mov_b32 m0, s76"]   # m0 saved / restored instead of clobbered (see gen_attn_w48.py)
    here = os.path.dirname(os.path.abspath(__file__))
    d = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc")
    base = "gemm_asm_hybrid_192x128" if BDIRECT else "gemm_asm_ring_192x128"
    with open(os.path.join(d, base + ("_stamps.inc" if STAMPS or ABLATE else ".inc")), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm_ring.py - do not edit. gfx950 assembly main loop of gemm_bf16_kernel_asm_ring (gemm.hip).\n")
        for ln in lines:
            f.write('"' + ln + '\\n\\t"\n')
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(MI * NI * 4)] + [f"s{i}" for i in range(36, 77)] + ["vcc", "scc", "memory"]
    with open(os.path.join(d, base + "_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_gemm_asm_ring.py - do not edit. Registers the assembly main loop assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    print(f"{len(lines)} lines, {sum(1 for ln in lines if 'v_mfma' in ln)} MFMAs")


if __name__ == "__main__":
    main()
