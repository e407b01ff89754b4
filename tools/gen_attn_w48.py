#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/attention_w48_asm.inc: the gfx950 assembly body of the 48-queries-per-wave attention
kernel (attention.hip, attn_fwd_kernel_w48_asm). The generated file is committed; the build does not run this script.

Layout, LDS images and the MFMA operand mapping are those of attn_fwd_kernel_w48_ref (plain HIP, same file) - that kernel pins
them through the parity tests. What the assembly adds is the part the compiler could not hold (Attention):
one wave per SIMD with hand-assigned registers (O in AGPRs, double-buffered S, Q and P in VGPRs) and one instruction stream in which
the softmax of tile t sits between the K.Q^T MFMAs of tile t+1, and the LDS-DMA of tile t+3 between the P.V MFMAs of tile t.

Register map (per wave):
  v[0:47]     Q fragments  qf[qb][ks]            a[0:95]   O accumulators o[db][qb] (4 each)
  v[48:95]    S buffer A   s[kb][qb] (4 each)    s[36:39]  K buffer descriptor      s[40:43] Vt buffer descriptor
  v[96:143]   S buffer B                         s44 / s45 K / Vt scalar offset of the next tile to stage
  v[144:167]  P fragments  pf[qb][i]             s46       loop counter (groups of 4 tiles)
  v[168:183]  fragment ring (4 x 4)              s[48:53]  compare masks of the rescale test
  v[184:207]  softmax temporaries (8 per query block)
  v[208:219]  m_ref[3], m_thr[3], -m_ref*c [3], l[3]   v[220:225] fragment addresses + 64 KB (ring slots 2, 3)
Operands (compiler-assigned): see the asm statement in attention.hip.
"""
import os

QF, SA, SB, PF, RING, TMP = 0, 48, 96, 144, 168, 184
MREF, MTHR, NMC, LRUN, HI = 208, 211, 214, 217, 220   # per query block: reference max, rescale threshold, -m_ref*c, row-sum partial
TAU = 8.0  # rescale only when a score exceeds the reference maximum by more than 2^TAU (after the scale): P stays <= 256
STAGE = 32768
RA = 3  # fragment reads in flight ahead of their MFMAs


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


def s_reg(buf, kb, qb, j=None):
    b = buf + (kb * 3 + qb) * 4
    return b if j is None else b + j


def o_reg(db, qb):
    return (db * 3 + qb) * 4


def kblock_off(kb):
    return (32 * (kb >> 1) + 4 * (kb & 1)) * 256


def vblock_off(db):
    return db * 16 * 128


class Gen:
    def __init__(self, stamps=False):
        self.lines = []
        self.stamps = stamps

    def stamp(self, i):
        """--stamps build only: s_memtime into s[54+2i : 55+2i] (written out at the end of the kernel)."""
        if self.stamps:
            self.e(f"s_memtime s[{54 + 2 * i}:{55 + 2 * i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    def e(self, s):
        self.lines.append(s)

    # ---- fragment reads: slot 0/1 use the operand addresses, slot 2/3 the +64 KB copies ----
    def k_read(self, ring, slot, kb, ks):
        addr = f"%[ka{ks}]" if slot < 2 else f"v{HI + ks}"
        return f"ds_read_b128 {vr(RING + 4 * ring)}, {addr} offset:{(slot & 1) * STAGE + kblock_off(kb)}"

    def v_read(self, ring, slot, db, i):
        addr = f"%[va{i}]" if slot < 2 else f"v{HI + 4 + i}"
        return f"ds_read_b128 {vr(RING + 4 * ring)}, {addr} offset:{(slot & 1) * STAGE + vblock_off(db)}"

    # ---- MFMA streams: list of groups, each group = [pre-instructions..., 3 MFMAs] per fragment ----
    def qk_stream(self, sbuf, slot):
        """S(next) = K Q^T from ring slot `slot` into S buffer `sbuf`. Returns a list of (kind, text)."""
        out = []
        frags = [(kb, ks) for kb in range(4) for ks in range(4)]
        issued = 0
        for f in range(min(RA, 16)):
            out.append(("ds", self.k_read(f % 4, slot, *frags[f])))
            issued += 1
        for f, (kb, ks) in enumerate(frags):
            out.append(("wait", f"s_waitcnt lgkmcnt({issued - f - 1})"))
            for qb in range(3):
                d = vr(s_reg(sbuf, kb, qb))
                c = "0" if ks == 0 else d
                out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {d}, {vr(RING + 4 * (f % 4))}, {vr(QF + (qb * 4 + ks) * 4)}, {c}"))
            if f + RA < 16:
                out.append(("ds", self.k_read((f + RA) % 4, slot, *frags[f + RA])))
                issued += 1
        return out

    def pv_stream(self, slot):
        out = []
        frags = [(db, i) for db in range(8) for i in range(2)]
        issued = 0
        for f in range(RA):
            out.append(("ds", self.v_read(f % 4, slot, *frags[f])))
            issued += 1
        for f, (db, i) in enumerate(frags):
            out.append(("wait", f"s_waitcnt lgkmcnt({issued - f - 1})"))
            for qb in range(3):
                a = f"a[{o_reg(db, qb)}:{o_reg(db, qb) + 3}]"
                out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {a}, {vr(RING + 4 * (f % 4))}, {vr(PF + (qb * 2 + i) * 4)}, {a}"))
            if f + RA < 16:
                out.append(("ds", self.v_read((f + RA) % 4, slot, *frags[f + RA])))
                issued += 1
        return out

    # ---- softmax of one S buffer (in place) ----
    # The O rescale costs 288 instructions for a wave's 96 accumulator registers (AGPR -> VGPR -> multiply -> AGPR), and with 48
    # queries per wave SOME query's maximum moves in nearly every tile. So the running maximum is replaced by a REFERENCE maximum
    # m_ref per query that is only raised when a score exceeds it by more than TAU (in exp2 units): P = exp2(s*c - m_ref*c) then
    # stays <= 2^TAU, l accumulates against the same reference, and the common path needs no alpha, no cross-lane reduction and
    # no rescale. The rare path (always taken on the first tile, m_ref = -inf) updates every query to its true running maximum.
    def sm_max(self, sbuf):
        """Phase 1: per-lane maximum of each query block's 16 scores and the wave-wide 'exceeds threshold' masks."""
        lists = []
        for qb in range(3):
            t = TMP + 8 * qb
            vals = [s_reg(sbuf, kb, qb, j) for kb in range(4) for j in range(4)]
            out = [f"v_max3_f32 v{t}, v{vals[0]}, v{vals[1]}, v{vals[2]}"]
            k = 3
            while k < 16:
                if k + 1 < 16:
                    out.append(f"v_max3_f32 v{t}, v{t}, v{vals[k]}, v{vals[k + 1]}")
                    k += 2
                else:
                    out.append(f"v_max_f32 v{t}, v{t}, v{vals[k]}")
                    k += 1
            out.append(f"v_cmp_gt_f32_e64 s[{48 + 2 * qb}:{49 + 2 * qb}], v{t}, v{MTHR + qb}")
            lists.append(out)
        return [x[k] for k in range(len(lists[0])) for x in lists]

    def sm_exp(self, sbuf):
        """Phase 2: P = exp2(s*c - m_ref*c), row-sum partials, bf16 pack into the PV product's B operand."""
        lists = []
        for qb in range(3):
            t = TMP + 8 * qb
            vals = [s_reg(sbuf, kb, qb, j) for kb in range(4) for j in range(4)]
            out = []
            for v in vals:
                out.append(f"v_fma_f32 v{v}, v{v}, %[c], v{NMC + qb}")
            for v in vals:
                out.append(f"v_exp_f32 v{v}, v{v}")
            tup = [s_reg(sbuf, kb, qb) for kb in range(4)]
            out.append(f"v_pk_add_f32 v[{t + 4}:{t + 5}], v[{tup[0]}:{tup[0] + 1}], v[{tup[0] + 2}:{tup[0] + 3}]")
            for b in tup[1:]:
                out.append(f"v_pk_add_f32 v[{t + 4}:{t + 5}], v[{t + 4}:{t + 5}], v[{b}:{b + 1}]")
                out.append(f"v_pk_add_f32 v[{t + 4}:{t + 5}], v[{t + 4}:{t + 5}], v[{b + 2}:{b + 3}]")
            out.append(f"v_add_f32 v{t + 4}, v{t + 4}, v{t + 5}")
            out.append(f"v_add_f32 v{LRUN + qb}, v{LRUN + qb}, v{t + 4}")
            # k-step i = key blocks 2i (low 4 values) and 2i+1 (high 4)
            for i in range(2):
                p = PF + (qb * 2 + i) * 4
                for hh in range(2):
                    b = s_reg(sbuf, 2 * i + hh, qb)
                    out.append(f"v_cvt_pk_bf16_f32 v{p + 2 * hh}, v{b}, v{b + 1}")
                    out.append(f"v_cvt_pk_bf16_f32 v{p + 2 * hh + 1}, v{b + 2}, v{b + 3}")
            lists.append(out)
        return [x[k] for k in range(len(lists[0])) for x in lists]

    def sm_check_and_rare_path(self, label):
        """After phase 1: if any lane saw a score above its threshold, raise every query's reference to its running maximum and
        rescale l and O accordingly (v{TMP+8qb} holds the lane's maximum of this tile)."""
        e = self.e
        e("s_nop 3")
        e("s_or_b64 s[48:49], s[48:49], s[50:51]")
        e("s_or_b64 s[48:49], s[48:49], s[52:53]")
        e("s_cmp_lg_u64 s[48:49], 0")
        e(f"s_cbranch_scc0 {label}f")
        for qb in range(3):
            t = TMP + 8 * qb
            for swap in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):  # maximum over the four lanes of a query
                e(f"v_mov_b32 v{t + 1}, v{t}")
                e("s_nop 1")
                e(f"{swap} v{t}, v{t + 1}")
                e("s_nop 1")
                e(f"v_max_f32 v{t}, v{t}, v{t + 1}")
            e(f"v_max_f32 v{t + 2}, v{MREF + qb}, v{t}")            # new reference = running maximum
            e(f"v_sub_f32 v{t + 3}, v{MREF + qb}, v{t + 2}")        # <= 0, -inf on the first tile
            e(f"v_mul_f32 v{t + 3}, %[c], v{t + 3}")
            e(f"v_exp_f32 v{t + 3}, v{t + 3}")                        # alpha
            e(f"v_mov_b32 v{MREF + qb}, v{t + 2}")
            e(f"v_add_f32 v{MTHR + qb}, %[tauc], v{t + 2}")          # threshold = reference + TAU / c
            e(f"v_mul_f32 v{NMC + qb}, %[c], v{t + 2}")
            e(f"v_sub_f32 v{NMC + qb}, 0, v{NMC + qb}")              # -m_ref * c
            e(f"v_mul_f32 v{LRUN + qb}, v{LRUN + qb}, v{t + 3}")
            e("s_nop 7")
            e("s_nop 7")
            for db in range(8):
                a = o_reg(db, qb)
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{t + 4 + j}, a{a + j}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_mul_f32 v{t + 4 + j}, v{t + 4 + j}, v{t + 3}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_accvgpr_write_b32 a{a + j}, v{t + 4 + j}")
        e("s_nop 7")
        e(f"{label}:")

    def interleave(self, stream, valu, per_mfma):
        """Emit (part of) an MFMA stream with `per_mfma` instructions of `valu` after every MFMA; leftovers at the end."""
        vi = 0
        for kind, text in stream:
            self.e(text)
            if kind == "mfma":
                for _ in range(per_mfma):
                    if vi < len(valu):
                        self.e(valu[vi])
                        vi += 1
        while vi < len(valu):
            self.e(valu[vi])
            vi += 1

    @staticmethod
    def split_stream(stream, n_mfma):
        """Cut an MFMA stream after its n_mfma-th MFMA."""
        k = 0
        for idx, (kind, _) in enumerate(stream):
            if kind == "mfma":
                k += 1
                if k == n_mfma:
                    return stream[:idx + 1], stream[idx + 1:]
        return stream, []

    def stage(self, slot):
        """LDS-DMA of the tile at scalar offsets s44 / s45 into ring slot `slot`; returns instruction pairs for interleaving."""
        out = []
        for i in range(4):
            out.append([f"s_add_u32 m0, %[wlds], {slot * STAGE + i * 4096}", "s_nop 0",
                        f"buffer_load_dwordx4 %[ko{i}], s[36:39], s44 offen lds"])
        for i in range(4):
            out.append([f"s_add_u32 m0, %[wlds], {slot * STAGE + 16384 + i * 4096}", "s_nop 0",
                        f"buffer_load_dwordx4 %[vo{i}], s[40:43], s45 offen lds"])
        return out

    def advance_stage_offsets(self):
        self.e("s_add_u32 s44, s44, %[ktb]")
        self.e("s_add_u32 s45, s45, 128")

    def step(self, slot, cur, nxt, uid):
        self.e(f"; ---------------- tile step, ring slot {slot} ----------------")
        st = self.stamp if slot == 0 else (lambda i: None)
        st(0)
        # part A: S(t+1) from slot+1; in the gaps first the per-lane maxima of S(t) + the (rarely taken) reference update,
        # then the exponentials, row sums and the bf16 pack
        qk = self.qk_stream(nxt, (slot + 1) & 3)
        head, tail = self.split_stream(qk, 6)
        self.interleave(head, self.sm_max(cur), 5)
        st(1)
        self.sm_check_and_rare_path(f"{uid}")
        st(2)
        self.interleave(tail, self.sm_exp(cur), 4)
        # part B: O += Vt P with the LDS-DMA of tile t+3 in the gaps
        dma = self.stage((slot + 3) & 3)
        stream = self.pv_stream(slot)
        nm = 0
        for kind, text in stream:
            self.e(text)
            if kind == "mfma":
                nm += 1
                if nm % 6 == 3 and dma:
                    for ins in dma.pop(0):
                        self.e(ins)
        for grp in dma:
            for ins in grp:
                self.e(ins)
        self.advance_stage_offsets()
        st(3)
        self.e("s_waitcnt vmcnt(8)")
        self.e("s_barrier")
        st(4)

    def build(self):
        e = self.e
        e("; ---- descriptors, constants ----")
        e("s_mov_b32 s36, %[kblo]")
        e("s_mov_b32 s37, %[kbhi]")
        e("s_mov_b32 s38, %[krec]")
        e("s_mov_b32 s39, 0x00020000")
        e("s_mov_b32 s40, %[vblo]")
        e("s_mov_b32 s41, %[vbhi]")
        e("s_mov_b32 s42, %[vrec]")
        e("s_mov_b32 s43, 0x00020000")
        e("s_mov_b32 s44, 0")
        e("s_mov_b32 s45, 0")
        e("s_mov_b32 s46, %[nt4]")
        for ks in range(4):
            e(f"v_add_u32 v{HI + ks}, 0x10000, %[ka{ks}]")
        for i in range(2):
            e(f"v_add_u32 v{HI + 4 + i}, 0x10000, %[va{i}]")
        e("; ---- Q fragments ----")
        for qb in range(3):
            for ks in range(4):
                e(f"global_load_dwordx4 {vr(QF + (qb * 4 + ks) * 4)}, %[qo{qb}], %[qbase] offset:{ks * 64}")
        for i in range(96):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        for qb in range(3):
            e(f"v_mov_b32 v{MREF + qb}, 0xff800000")
            e(f"v_mov_b32 v{MTHR + qb}, 0xff800000")
            e(f"v_mov_b32 v{NMC + qb}, 0")
            e(f"v_mov_b32 v{LRUN + qb}, 0")
        e("; ---- tiles 0, 1, 2 ----")
        for t in range(3):
            for grp in self.stage(t):
                for ins in grp:
                    e(ins)
            self.advance_stage_offsets()
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        for kind, text in self.qk_stream(SA, 0):
            e(text)
        e("s_nop 7")
        e("s_nop 7")
        e("10:")
        self.step(0, SA, SB, 11)
        self.step(1, SB, SA, 12)
        self.step(2, SA, SB, 13)
        self.step(3, SB, SA, 14)
        e("s_sub_u32 s46, s46, 1")
        e("s_cmp_lg_u32 s46, 0")
        e("s_cbranch_scc1 10b")
        e("; ---- epilogue ----")
        e("s_waitcnt vmcnt(0)")
        e("s_nop 7")
        e("s_nop 7")
        for qb in range(3):
            t = TMP + 8 * qb
            e(f"v_mov_b32 v{t}, v{LRUN + qb}")
            for swap in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
                e(f"v_mov_b32 v{t + 1}, v{t}")
                e("s_nop 1")
                e(f"{swap} v{t}, v{t + 1}")
                e("s_nop 1")
                e(f"v_add_f32 v{t}, v{t}, v{t + 1}")
            e(f"v_rcp_f32 v{t}, v{t}")
            e("s_nop 1")
            for db in range(8):
                a = o_reg(db, qb)
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{t + 2 + j}, a{a + j}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_mul_f32 v{t + 2 + j}, v{t + 2 + j}, v{t}")
                e(f"v_cvt_pk_bf16_f32 v{t + 6}, v{t + 2}, v{t + 3}")
                e(f"v_cvt_pk_bf16_f32 v{t + 7}, v{t + 4}, v{t + 5}")
                e(f"global_store_dwordx2 %[oo{qb}], v[{t + 6}:{t + 7}], %[obase] offset:{db * 32}")
        e("s_waitcnt vmcnt(0)")
        if self.stamps:  # lane 0 of every wave writes its 5 stamps: dbg[wave][5] u64
            for i in range(10):
                e(f"v_mov_b32 v{TMP + i}, s{54 + i}")
            e(f"v_mov_b32 v{TMP + 10}, 0")
            for i in range(5):
                e(f"global_store_dwordx2 v{TMP + 10}, v[{TMP + 2 * i}:{TMP + 2 * i + 1}], %[dbg] offset:{i * 8}")
            e("s_waitcnt vmcnt(0)")
        return self.lines


def main():
    import sys
    stamps = "--stamps" in sys.argv
    g = Gen(stamps)
    lines = g.build()
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc", "attention_w48_asm_stamps.inc" if stamps else "attention_w48_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_w48.py - do not edit. gfx950 assembly body of attn_fwd_kernel_w48_asm (attention.hip).\n")
        for ln in lines:
            f.write('"' + ln.replace('"', '\\"') + '\\n\\t"\n')
    clob = [f"v{i}" for i in range(230)] + [f"a{i}" for i in range(96)] + [f"s{i}" for i in range(36, 64)] + ["m0", "vcc", "scc", "memory"]
    with open(os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc", "attention_w48_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_w48.py - do not edit. Registers the assembly body assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    n_mfma = sum(1 for ln in lines if "v_mfma" in ln)
    print(f"{len(lines)} lines, {n_mfma} MFMAs -> {os.path.normpath(out)}")


if __name__ == "__main__":
    main()
