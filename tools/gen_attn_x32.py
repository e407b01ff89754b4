#!/usr/bin/env python3
"""Generates ltx-video-swift-mlx_amd/csrc/attention_x32_asm.inc: the gfx950 assembly body of attn_fwd_kernel_x32_asm (attention.hip),
the attention kernel on v_mfma_f32_32x32x16_bf16 with 48 queries per wave. The generated file is committed; the build does not run
this script.

Why this shape ("Attention in round 3"): the 16x16x32 stream of gen_attn_w48.py needs 972 vector-issue cycles
per 64-key tile (48 exponentials x 16 + packs + maxima) and its 102 MFMAs leave 816 free - a 16-cycle MFMA hides 8 cycles of
VALU issue, a 32-cycle MFMA hides 24. 64 queries per wave on 32x32 blocks would fill only 768 of the 1024 SIMDs at 1536 tokens x 32
heads, so a wave here owns ONE 32-query block outright and HALF of a second one:

  workgroup = 192 queries = six 32-query blocks, 4 waves (one per SIMD). Wave w owns block w. Blocks 4 and 5 are shared by the
  wave pairs (0,1) and (2,3): of every 64-key tile, wave w takes the 32 keys of half hw = w & 1 for the shared block. Per tile a
  wave issues 16 + 8 K.Q^T MFMAs and 16 + 8 V^T.P MFMAs = 48 x 32 cycles = 1536 (the w48 stream: 102 x 16 = 1632), and every K / V^T
  fragment of its own half feeds two MFMAs. Each wave keeps its own reference maximum and row sum for its half of the shared block;
  the two partial (m, l, O) are combined once, through LDS, in the epilogue (each wave finalises two of the four d-blocks).

Everything else follows the w48 kernel: S^T = K.Q^T so that a lane owns a query column, P stays in registers as the PV product's B
operand through the key permutation applied when the K fragment is addressed (MFMA row rho reads key rho with bits 2 and 3 swapped),
scores arrive relative to a reference maximum from the MFMA itself (accumulator init = -m_ref), the reference is raised only when a
score exceeds it by 2^8 (rare path), Q is prescaled by scale * log2(e) by its producer (scores are base-2 exponents; no multiplies).
Row sums are f32 adds on the un-rounded P (a block of ones would cost a whole 32-cycle MFMA per k-step here).

A/B halves: "A" is the wave's own key half (keys 32 hw .. 32 hw + 31 of a tile), "B" the other one. The stream is the same for all
four waves; which half is A is folded into the per-wave fragment addresses (operands kbA / kbB, vaA* / vaB*).

Register map (per wave):
  a[0:63]     O own block   o[db] (16 each)         v[0:47]    S buffer 0: own A (16), own B (16), shared (16)
  a[64:127]   O shared block                        v[48:95]   S buffer 1
  a[128:159]  Q fragments own  qf[ks] (4 each)      v[96:119]  P fragments: own A [0,1], own B [0,1], shared [0,1] (4 each)
  a[160:191]  Q fragments shared                    v[120:135] -(reference max) own block, all 16 registers (accumulator init)
  a[192:223]  K / V^T fragment ring (8 x 4)         v[136:151] -(reference max) shared block
  v[152:183]  K fragment addresses: A lo, B lo, A hi, B hi (8 k-steps each; hi = + 64 KB: ring slots 2, 3)
  v[184:191]  V^T fragment addresses: A0 A1 B0 B1 lo, then hi
  v192 / v193 per-lane maxima own / shared, v[194:201] temporaries, v202 / v203 floor own / shared, v204 / v205 row sums own / shared
  s[36:39] K descriptor, s[40:43] V^T descriptor, s44 / s45 scalar offsets of the next tile to stage, s46 tiles left,
  s[48:51] compare masks, s[60:63] O descriptor
LDS: four ring slots of 32 KB ([64 keys][256 B] K image + [128 d][128 B] V^T image, both XOR-swizzled on the SOURCE address of the
LDS-DMA); the exchange area of the epilogue reuses slots 0 and 1.

Pipeline of one tile step t (S(t) is in the `cur` buffer, already checked against the reference):
  part A : S(t+1) = K(t+1).Q^T, 24 MFMAs over 16 fragments; fillers: P = exp2(S(t)) of own A and shared, row sums, bf16 packs, LDS-DMA
  part B1: O += V^T(t).P over the wave's own key half, 16 MFMAs over 8 fragments (own + shared block); fillers: P of own B, LDS-DMA
  part B2: O += V^T(t).P over the other half, 8 MFMAs over 8 fragments; fillers: per-lane maxima of S(t+1)
  then the reference check of S(t+1) (rare path: rescale O and l), `s_waitcnt vmcnt(0)`, barrier.
Fragment reads run seven ahead of their MFMAs ACROSS the step boundary: the first seven K reads of step t+1 are issued under part
B2 of step t. That is legal because the LDS-DMA of tile t+3 is issued early in step t and the step ends with vmcnt(0) + barrier, so
at every barrier the three tiles ahead are visible (check_wait_coverage proves it on the emitted text).
"""
import os
import re
import sys

# ---- accumulator file ----
O_OWN, O_SH, Q_OWN, Q_SH, ARING = 0, 64, 128, 160, 192
RN = 8            # ring entries (4 registers each)
RA = RN - 1       # fragment reads in flight ahead of their MFMAs
NA = ARING + 4 * RN
# ---- vector registers ----
SBUF = (0, 48)    # S buffers; inside one: own A +0, own B +16, shared +32
OWN_A, OWN_B, SH = 0, 16, 32
PF = 96           # pf[blk][s2]: own A 96 / 100, own B 104 / 108, shared 112 / 116
PF_OWN_A, PF_OWN_B, PF_SH = 96, 104, 112
NM_OWN, NM_SH = 120, 136
KA_LO, KB_LO, KA_HI, KB_HI = 152, 160, 168, 176
VA_LO, VA_HI = 184, 188   # A0 A1 B0 B1
MAXR_OWN, MAXR_SH = 192, 193
RT = 194          # v[194:201]
FLOOR_OWN, FLOOR_SH = 202, 203
L_OWN, L_SH = 204, 205
NV = 208
STAGE = 32768
V_OFF = 16384     # V^T image inside a slot (part of the vaXX operands)
STAGE_OPS = 8
TAU = 8.0
XSZ = 0x2400      # exchange area per wave: 8 KB of accumulators + 1 KB of (m, l)

ABL = set(os.environ.get("X32_ABLATE", "").split(","))  # timing experiments only (wrong results): nodma, noexp, nomax
GAP = float(os.environ.get("X32_GAP", "24"))           # vector-issue cycles a 32-cycle MFMA leaves free (guide: 8 of 32 are its own)
DMA_COST = float(os.environ.get("X32_DMA_COST", "28"))


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=4):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


def ring(i):
    return ar(ARING + 4 * (i % RN))


class Gen:
    def __init__(self, stamps=False):
        self.lines = []
        self.stamps = stamps

    def e(self, s):
        self.lines.append(s)

    # ------------------------------------------------------------------------------------------------------------
    # fragment reads
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def k_read(ri, slot, half, ks):
        base = (KA_LO, KB_LO, KA_HI, KB_HI)[(slot >> 1) * 2 + half]
        return f"ds_read_b128 {ring(ri)}, v{base + ks} offset:{(slot & 1) * STAGE}"

    @staticmethod
    def v_read(ri, slot, half, s2, db):
        base = (VA_LO, VA_HI)[slot >> 1]
        return f"ds_read_b128 {ring(ri)}, v{base + half * 2 + s2} offset:{(slot & 1) * STAGE + db * 4096}"

    def step_frags(self, nxt, kslot, vslot):
        """The 32 fragments of one step with their MFMAs. Each entry: (read text builder(ring index), [mfma texts builder(ring index)])."""
        fr = []
        for ks in range(8):   # k-step outer: the three accumulation chains of a k-step are independent
            for half in (0, 1):
                def rd(ri, half=half, ks=ks):
                    return self.k_read(ri, kslot, half, ks)

                def mf(ri, half=half, ks=ks):
                    out = []
                    blocks = ((OWN_A, Q_OWN, NM_OWN), (SH, Q_SH, NM_SH)) if half == 0 else ((OWN_B, Q_OWN, NM_OWN),)
                    for so, q, nm in blocks:
                        d = vr(nxt + so, 16)
                        c = vr(nm, 16) if ks == 0 else d   # scores arrive as q.k - m_ref
                        out.append(f"v_mfma_f32_32x32x16_bf16 {d}, {ring(ri)}, {ar(q + 4 * ks)}, {c}")
                    return out
                fr.append(("k", rd, mf))
        for half in (0, 1):
            for s2 in range(2):
                for db in range(4):
                    def rd(ri, half=half, s2=s2, db=db):
                        return self.v_read(ri, vslot, half, s2, db)

                    def mf(ri, half=half, s2=s2, db=db):
                        o = ar(O_OWN + 16 * db, 16)
                        out = [f"v_mfma_f32_32x32x16_bf16 {o}, {ring(ri)}, {vr((PF_OWN_A, PF_OWN_B)[half] + 4 * s2)}, {o}"]
                        if half == 0:
                            o = ar(O_SH + 16 * db, 16)
                            out.append(f"v_mfma_f32_32x32x16_bf16 {o}, {ring(ri)}, {vr(PF_SH + 4 * s2)}, {o}")
                        return out
                    fr.append(("v", rd, mf))
        return fr

    def step_stream(self, nxt, slot):
        """[(kind, text)] of one step: S(t+1) from K slot (slot+1)&3, O += V^T.P from slot `slot`, and - behind the last seven
        fragments - the first seven K reads of the NEXT step (K slot (slot+2)&3). The step starts with seven reads in flight."""
        fr = self.step_frags(nxt, (slot + 1) & 3, slot)
        nfr = self.step_frags(0, (slot + 2) & 3, (slot + 1) & 3)   # only its first RA read builders are used
        out = []
        for F, (kind, rd, mf) in enumerate(fr):
            out.append(("wait", f"s_waitcnt lgkmcnt({RA - 1})"))
            for m in mf(F):
                out.append(("mfma", m))
            G = F + RA
            out.append(("ds", fr[G][1](G) if G < len(fr) else nfr[G - len(fr)][1](G)))   # ring index continues: 32 % 8 == 0
        return out

    def first_reads(self, slot):
        """The seven K reads a step expects in flight at its start (prologue)."""
        fr = self.step_frags(0, slot, slot)
        return [fr[F][1](F) for F in range(RA)]

    # ------------------------------------------------------------------------------------------------------------
    # softmax pieces (all in place on an S buffer)
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def sm_exp_groups(buf, blk_off, pf, lreg):
        """P = exp2(S') for one 32-key block of one query block (16 registers -> pf[0], pf[1]), row-sum adds, bf16 packs.
        Returned as a list of single-instruction groups ordered so that no instruction uses a result produced by the one before it:
        exponentials of k-step 1 sit between the exponentials of k-step 0 and their adds / packs."""
        if "noexp" in ABL:
            return []
        r = buf + blk_off
        ex = [[f"v_exp_f32 v{r + j}, v{r + j}" for j in range(8 * s2, 8 * s2 + 8)] for s2 in range(2)]
        post = []
        for s2 in range(2):
            p = []
            for j in range(8 * s2, 8 * s2 + 8, 2):
                p.append(f"v_add_f32 v{lreg}, v{lreg}, v{r + j}")
                p.append(f"v_add_f32 v{lreg}, v{lreg}, v{r + j + 1}")
                p.append(f"v_cvt_pk_bf16_f32 v{pf + 4 * s2 + (j - 8 * s2) // 2}, v{r + j}, v{r + j + 1}")
            post.append(p)
        seq = list(ex[0])
        a, b = ex[1], post[0]
        while a or b:     # 8 exponentials against 12 adds / packs
            if a:
                seq.append(a.pop(0))
            for _ in range(2):
                if b:
                    seq.append(b.pop(0))
        seq += post[1]
        return [[x] for x in seq]

    @staticmethod
    def sm_max_groups(buf):
        """Per-lane maximum of the next tile's scores: own block (32 registers) and shared block (16), then the threshold compares."""
        if "nomax" in ABL:
            return [[f"v_mov_b32 v{MAXR_OWN}, 0"], [f"v_mov_b32 v{MAXR_SH}, 0"],
                    [f"v_cmp_gt_f32_e64 s[48:49], v{MAXR_OWN}, %[tau]"], [f"v_cmp_gt_f32_e64 s[50:51], v{MAXR_SH}, %[tau]"]]
        chains = []
        for t, regs in ((MAXR_SH, [buf + SH + j for j in range(16)]), (MAXR_OWN, [buf + j for j in range(32)])):
            out = [f"v_max3_f32 v{t}, v{regs[0]}, v{regs[1]}, v{regs[2]}"]
            k = 3
            while k < len(regs):
                if k + 1 < len(regs):
                    out.append(f"v_max3_f32 v{t}, v{t}, v{regs[k]}, v{regs[k + 1]}")
                    k += 2
                else:
                    out.append(f"v_max_f32 v{t}, v{t}, v{regs[k]}")
                    k += 1
            chains.append(out)
        sh, own = chains
        seq = []
        while sh or own:   # the shared block first (its last MFMA is the older one), then alternate: independent neighbours
            if sh:
                seq.append(sh.pop(0))
            if own:
                seq.append(own.pop(0))
            if own and not sh:
                seq.append(own.pop(0))
        seq.append(f"v_cmp_gt_f32_e64 s[48:49], v{MAXR_OWN}, %[tau]")
        seq.append(f"v_cmp_gt_f32_e64 s[50:51], v{MAXR_SH}, %[tau]")
        return [[x] for x in seq]

    def rare_path(self, label, buf, force=False):
        """After the maxima of `buf`: if any lane saw a score above the threshold, raise every query's reference to its running maximum:
        delta = max(row maximum, floor) in shifted units; S' -= delta, accumulator init -= delta, O and l *= 2^-delta."""
        e = self.e
        e("s_nop 3")
        e("s_or_b64 s[48:49], s[48:49], s[50:51]")
        e("s_cmp_lg_u64 s[48:49], 0")
        if not force:
            e(f"s_cbranch_scc0 {label}f")
        for t, floor, nm, sregs, obase, lreg in ((MAXR_OWN, FLOOR_OWN, NM_OWN, [buf + j for j in range(32)], O_OWN, L_OWN),
                                                 (MAXR_SH, FLOOR_SH, NM_SH, [buf + SH + j for j in range(16)], O_SH, L_SH)):
            e(f"v_mov_b32 v{RT}, v{t}")            # maximum over the two lanes of a query
            e("s_nop 1")
            e(f"v_permlane32_swap_b32 v{t}, v{RT}")
            e("s_nop 1")
            e(f"v_max_f32 v{t}, v{t}, v{RT}")
            e(f"v_max_f32 v{RT + 1}, v{t}, v{floor}")      # delta
            e(f"v_sub_f32 v{RT + 2}, 0, v{RT + 1}")
            e(f"v_min_f32 v{RT + 2}, 0, v{RT + 2}")        # first tile: the reference may move DOWN (O = l = 0 then)
            e(f"v_exp_f32 v{RT + 2}, v{RT + 2}")
            for j in range(16):
                e(f"v_sub_f32 v{nm + j}, v{nm + j}, v{RT + 1}")
            for r in sregs:
                e(f"v_sub_f32 v{r}, v{r}, v{RT + 1}")
            if force:
                continue   # first tile: O and l are still zero
            e(f"v_mul_f32 v{lreg}, v{lreg}, v{RT + 2}")
            for k in range(3):
                e("s_nop 7")   # the step's last MFMAs wrote O: MFMA result -> v_accvgpr_read
            for a in range(obase, obase + 64, 4):
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{RT + 4 + j}, a{a + j}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_mul_f32 v{RT + 4 + j}, v{RT + 4 + j}, v{RT + 2}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_accvgpr_write_b32 a{a + j}, v{RT + 4 + j}")
        e(f"v_mov_b32 v{FLOOR_OWN}, 0")
        e(f"v_mov_b32 v{FLOOR_SH}, 0")
        e("s_nop 7")
        e(f"{label}:")

    # ------------------------------------------------------------------------------------------------------------
    # LDS-DMA
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def stage(slot):
        out = []
        for i in range(4):
            out.append([f"s_add_u32 m0, %[wlds], {slot * STAGE + i * 4096}", "s_nop 0",
                        f"buffer_load_dwordx4 %[ko{i}], s[36:39], s44 offen lds"])
        for i in range(4):
            out.append([f"s_add_u32 m0, %[wlds], {slot * STAGE + V_OFF + i * 4096}", "s_nop 0",
                        f"buffer_load_dwordx4 %[vo{i}], s[40:43], s45 offen lds"])
        assert len(out) == STAGE_OPS
        return out

    def advance_stage_offsets(self):
        self.e("s_add_u32 s44, s44, %[ktb]")
        self.e("s_add_u32 s45, s45, 128")

    # ------------------------------------------------------------------------------------------------------------
    # placement
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def cost(ins):
        if ins.startswith(("v_exp_f32", "v_rcp_f32")):
            return 8.0
        if ins.startswith("buffer_load"):
            return DMA_COST
        if ins.startswith("ds_read"):
            return 4.0
        if ins.startswith(("s_nop", "s_add", "s_cmp", "s_sub", "s_waitcnt")):
            return 1.0
        return 4.0

    def spread(self, stream, groups, skip=0):
        """Emit `stream`; filler groups go into the gaps behind its MFMAs (not behind the first `skip` ones) in proportion to what
        each gap has left of GAP vector-issue cycles after the fragment read / wait it already carries. Leftovers at the end."""
        items = list(stream)
        mf = [i for i, (k, _) in enumerate(items) if k == "mfma"]
        free = {}
        for n, i in enumerate(mf):
            if n < skip:
                continue
            j = i + 1
            base = 0.0
            while j < len(items) and items[j][0] != "mfma":
                base += self.cost(items[j][1])
                j += 1
            free[i] = max(0.0, GAP - base)
        demand = sum(sum(self.cost(x) for x in g) for g in groups)
        cap = sum(free.values())
        f = max(1.0, demand / max(cap, 1.0))
        gi, cum, allowed = 0, 0.0, 0.0
        for i, (kind, text) in enumerate(items):
            self.e(text)
            if i in free:
                allowed += free[i] * f
                while gi < len(groups) and cum + sum(self.cost(x) for x in groups[gi]) <= allowed + 0.5:
                    for ins in groups[gi]:
                        self.e(ins)
                    cum += sum(self.cost(x) for x in groups[gi])
                    gi += 1
        while gi < len(groups):
            for ins in groups[gi]:
                self.e(ins)
            gi += 1

    @staticmethod
    def split_stream(stream, n_mfma):
        """Cut behind the n_mfma-th MFMA, keeping the fragment read that follows it with the first part."""
        k = 0
        for idx, (kind, _) in enumerate(stream):
            if kind == "mfma":
                k += 1
                if k == n_mfma:
                    end = idx + 1
                    while end < len(stream) and stream[end][0] == "ds":
                        end += 1
                    return stream[:end], stream[end:]
        return stream, []

    def stamp(self, i):
        """--stamps build with X32_PART_STAMPS=1: part boundaries of the slot-0 step (each stamp drains the LDS queue: it distorts)."""
        if self.stamps and os.environ.get("X32_PART_STAMPS"):
            self.e(f"s_memtime s[{84 + 2 * i}:{85 + 2 * i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    def kstamp(self, i):
        """--stamps build: whole-kernel stamps (start, loop entered, loop left, stores issued) in s[76+2i : 77+2i]."""
        if self.stamps:
            self.e(f"s_memtime s[{76 + 2 * i}:{77 + 2 * i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    # ------------------------------------------------------------------------------------------------------------
    # one tile step
    # ------------------------------------------------------------------------------------------------------------
    def step_fillers(self, slot, cur, nxt):
        """Everything of a step that is not an MFMA or a fragment read, as (target gap, latest gap, [instructions]).
        Gap g = behind the step's g-th MFMA (0..47: 24 of K.Q^T, 16 of V^T.P over the own key half, 8 over the other half).

        One exponential per gap is what a 32-cycle MFMA hides beside a few cheap instructions (v_exp_f32 holds the transcendental
        unit for 16 cycles): the 48 exponentials of S(t) are laid over gaps 0..~41 in the order the PV product consumes them (own A
        k-step 0, shared 0, own A 1, shared 1, own B 0, own B 1), each pair's row-sum adds and bf16 pack one gap behind it. A P
        fragment must be complete before the first MFMA that reads it: that MFMA's index - 1 is the group's latest gap."""
        items = []
        squeeze = float(os.environ.get("X32_EXP_SQUEEZE", "0.875"))   # gaps per exponential
        groups = [(OWN_A, 0, PF_OWN_A, L_OWN, 23), (SH, 0, PF_SH, L_SH, 24), (OWN_A, 1, PF_OWN_A, L_OWN, 31), (SH, 1, PF_SH, L_SH, 32),
                  (OWN_B, 0, PF_OWN_B, L_OWN, 39), (OWN_B, 1, PF_OWN_B, L_OWN, 43)]
        if "noexp" not in ABL:
            i = 0
            for blk, s2, pf, lreg, last in groups:
                r = cur + blk + 8 * s2
                for j in range(0, 8, 2):
                    t0, t1 = i * squeeze, (i + 1) * squeeze
                    items.append((t0, last - 1, [f"v_exp_f32 v{r + j}, v{r + j}"]))
                    items.append((t1, last - 1, [f"v_exp_f32 v{r + j + 1}, v{r + j + 1}"]))
                    post = float(os.environ.get("X32_POST_LAG", "1.0"))
                    items.append((t1 + post, last, [f"v_add_f32 v{lreg}, v{lreg}, v{r + j}"]))
                    items.append((t1 + post + 0.01, last, [f"v_add_f32 v{lreg}, v{lreg}, v{r + j + 1}"]))
                    items.append((t1 + post + 0.02, last, [f"v_cvt_pk_bf16_f32 v{pf + 4 * s2 + j // 2}, v{r + j}, v{r + j + 1}"]))
                    i += 2
        if "nodma" not in ABL:
            d0, dstep = float(os.environ.get("X32_DMA_FIRST", "2")), float(os.environ.get("X32_DMA_STEP", "4"))
            for k, grp in enumerate(self.stage((slot + 3) & 3)):
                items.append((d0 + dstep * k + 0.5, 40, grp))
        mx = self.sm_max_groups(nxt)
        m0 = float(os.environ.get("X32_MAX_FIRST", "28"))   # S(t+1) is complete behind MFMA 23; four MFMAs of distance to its readers
        for k, g in enumerate(mx):
            items.append((max(28.0, m0 + k * (47.9 - m0) / len(mx)), 47, g))
        return items

    def step(self, slot, cur, nxt, uid):
        e = self.e
        e(f"; ---------------- tile step, ring slot {slot} ----------------")
        st = self.stamp if slot == 0 else (lambda i: None)
        st(0)
        stream = self.step_stream(nxt, slot)
        items = sorted(self.step_fillers(slot, cur, nxt), key=lambda it: it[0])
        by_gap = {}
        for t, last, grp in items:
            g = min(int(t), last, 47)
            by_gap.setdefault(g, []).append(grp)
        n = -1
        for kind, text in stream:
            e(text)
            if kind == "mfma":
                n += 1
                for grp in by_gap.get(n, []):
                    for ins in grp:
                        e(ins)
                if n == 23:
                    st(1)
                if n == 39:
                    st(2)
        assert n == 47
        # tile t was the last one: leave before the reference check of a tile that does not exist
        e("s_sub_u32 s46, s46, 1")
        e("s_cmp_eq_u32 s46, 0")
        e("s_cbranch_scc1 30f")
        self.rare_path(f"{uid}", nxt)
        self.advance_stage_offsets()
        st(3)
        # every LDS-DMA issued so far has landed: with the barrier, tiles t+1 .. t+3 are visible to every wave
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        st(4)

    # ------------------------------------------------------------------------------------------------------------
    # whole kernel body
    # ------------------------------------------------------------------------------------------------------------
    def build(self):
        e = self.e
        self.kstamp(0)
        e("; ---- descriptors, constants ----")
        for i, v in enumerate(("%[kblo]", "%[kbhi]", "%[krec]", "0x00020000")):
            e(f"s_mov_b32 s{36 + i}, {v}")
        for i, v in enumerate(("%[vblo]", "%[vbhi]", "%[vrec]", "0x00020000")):
            e(f"s_mov_b32 s{40 + i}, {v}")
        e("s_mov_b32 s44, 0")
        e("s_mov_b32 s45, 0")
        e("s_mov_b32 s46, %[nt]")
        for i, v in enumerate(("%[oblo]", "%[obhi]", "%[orec]", "0x00020000")):
            e(f"s_mov_b32 s{60 + i}, {v}")
        e("; ---- Q fragments (B operand of K.Q^T) straight into the accumulator file ----")
        for ks in range(8):
            e(f"global_load_dwordx4 {ar(Q_OWN + 4 * ks)}, %[qoo], %[qbase] offset:{ks * 32}")
        for ks in range(8):
            e(f"global_load_dwordx4 {ar(Q_SH + 4 * ks)}, %[qos], %[qbase] offset:{ks * 32}")
        e("; ---- tiles 0, 1, 2 ----")
        for t in range(3):
            for grp in self.stage(t):
                for ins in grp:
                    e(ins)
            self.advance_stage_offsets()
        e("; ---- fragment addresses: ((ks << 5) ^ kx5) + base; + 64 KB for ring slots 2, 3 ----")
        for ks in range(8):
            e(f"v_xor_b32 v{RT}, {ks << 5}, %[kx5]")
            e(f"v_add_u32 v{KA_LO + ks}, v{RT}, %[kbA]")
            e(f"v_add_u32 v{KB_LO + ks}, v{RT}, %[kbB]")
            e(f"v_add_u32 v{KA_HI + ks}, 0x10000, v{KA_LO + ks}")
            e(f"v_add_u32 v{KB_HI + ks}, 0x10000, v{KB_LO + ks}")
        for i, op in enumerate(("%[vaA0]", "%[vaA1]", "%[vaB0]", "%[vaB1]")):
            e(f"v_mov_b32 v{VA_LO + i}, {op}")
            e(f"v_add_u32 v{VA_HI + i}, 0x10000, {op}")
        for i in range(128):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        for j in range(16):
            e(f"v_mov_b32 v{NM_OWN + j}, 0")
            e(f"v_mov_b32 v{NM_SH + j}, 0")
        e(f"v_mov_b32 v{FLOOR_OWN}, 0xff800000")
        e(f"v_mov_b32 v{FLOOR_SH}, 0xff800000")
        e(f"v_mov_b32 v{L_OWN}, 0")
        e(f"v_mov_b32 v{L_SH}, 0")
        # issue order: Q (16 loads), K0 (4), V0 (4), tile 1 (8), tile 2 (8). The first K.Q^T needs Q and K0 only.
        e(f"s_waitcnt vmcnt({3 * STAGE_OPS - STAGE_OPS // 2})")
        e("s_barrier")
        e("; ---- S(0) = K(0).Q^T, synchronous ----")
        fr = self.step_frags(SBUF[0], 0, 0)[:16]
        for F in range(RA):
            e(fr[F][1](F))
        for F, (kind, rd, mf) in enumerate(fr):
            e(f"s_waitcnt lgkmcnt({min(RA, 16 - F) - 1})")
            for m in mf(F):
                e(m)
            if F + RA < 16:
                e(fr[F + RA][1](F + RA))
        for k in range(3):
            e("s_nop 7")
        for g in self.sm_max_groups(SBUF[0]):
            for ins in g:
                e(ins)
        self.rare_path("9", SBUF[0], force=True)
        e("s_waitcnt vmcnt(0)")     # tiles 0..2 have landed
        e("s_barrier")
        for r in self.first_reads(1):   # the first step's K fragments (tile 1), seven ahead
            e(r)
        self.kstamp(1)
        e("10:")
        if self.stamps:   # s[64:65] = start of this loop iteration (4 tile steps), s[66:67] = start of the previous one
            e("s_mov_b64 s[66:67], s[64:65]")
            e("s_memtime s[64:65]")
            e("s_waitcnt lgkmcnt(0)")
        self.step(0, SBUF[0], SBUF[1], 11)
        self.step(1, SBUF[1], SBUF[0], 12)
        self.step(2, SBUF[0], SBUF[1], 13)
        self.step(3, SBUF[1], SBUF[0], 14)
        e("s_branch 10b")
        e("30:")
        self.kstamp(2)
        self.epilogue()
        if self.stamps:
            e("s_waitcnt vmcnt(0)")
            self.kstamp(3)
            T = SBUF[0]
            for i in range(4):     # s[64:67] loop-iteration stamps
                e(f"v_mov_b32 v{T + i}, s{64 + i}")
            for i in range(8):     # s[76:83] kernel stamps
                e(f"v_mov_b32 v{T + 4 + i}, s{76 + i}")
            for i in range(10):    # s[84:93] part stamps
                e(f"v_mov_b32 v{T + 12 + i}, s{84 + i}")
            e(f"v_mov_b32 v{T + 24}, 0")
            for i in range(11):
                e(f"global_store_dwordx2 v{T + 24}, v[{T + 2 * i}:{T + 2 * i + 1}], %[dbg] offset:{i * 8}")
            e("s_waitcnt vmcnt(0)")
        return self.lines

    def epilogue(self):
        e = self.e
        e("; ---- epilogue ----")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")   # LDS-DMA issued past the last tile, fragment reads issued ahead
        e("s_barrier")                        # every wave has left the ring: slots 0, 1 become the exchange area
        for k in range(3):
            e("s_nop 7")
        T = SBUF[0]   # v[0:95] are free now
        # ---- shared block: publish (m, l) and the two d-blocks the partner finalises ----
        e(f"v_mov_b32 v{RT}, v{L_SH}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 v{L_SH}, v{RT}")
        e("s_nop 1")
        e(f"v_add_f32 v{L_SH}, v{L_SH}, v{RT}")            # l of this wave's half, both lanes of a query
        e(f"v_sub_f32 v{T + 64}, 0, v{NM_SH}")              # m
        e(f"v_mov_b32 v{T + 65}, v{L_SH}")
        e(f"v_mov_b32 v{T + 66}, 0")
        e(f"v_mov_b32 v{T + 67}, 0")
        e(f"ds_write_b128 %[xout], v[{T + 64}:{T + 67}] offset:8192")
        e("s_cmp_eq_u32 %[hw], 0")
        e("s_cbranch_scc0 41f")
        self.shared_finish(keep=(0, 1), send=(2, 3), T=T)
        e("s_branch 42f")
        e("41:")
        self.shared_finish(keep=(2, 3), send=(0, 1), T=T)
        e("42:")
        # ---- own block ----
        e(f"v_mov_b32 v{RT}, v{L_OWN}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 v{L_OWN}, v{RT}")
        e("s_nop 1")
        e(f"v_add_f32 v{L_OWN}, v{L_OWN}, v{RT}")
        e("s_nop 0")
        e(f"v_rcp_f32 v{RT}, v{L_OWN}")
        e("s_nop 1")
        n = 0
        for db in range(4):
            for g4 in range(4):
                t = T + 8 * (n % 12)   # rotating temporaries: a store's data registers are not rewritten for 11 groups
                n += 1
                a = O_OWN + 16 * db + 4 * g4
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{t + j}, a{a + j}")
                e("s_nop 0")
                for j in range(4):
                    e(f"v_mul_f32 v{t + j}, v{t + j}, v{RT}")
                e(f"v_cvt_pk_bf16_f32 v{t + 4}, v{t}, v{t + 1}")
                e(f"v_cvt_pk_bf16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
                e(f"buffer_store_dwordx2 v[{t + 4}:{t + 5}], %[oow], s[60:63], 0 offen offset:{db * 64 + g4 * 16}")
        # no wait for the stores: they may complete after the wave ends

    def shared_finish(self, keep, send, T):
        """One wave's half of the shared-block combine. `send`: the d-blocks whose partial sums go to the partner (8 x 16 bytes per
        lane), `keep`: the ones finalised here with the partner's partial sums. v[T : T+95] are temporaries."""
        e = self.e
        q = 0
        for db in send:
            for g4 in range(4):
                a = O_SH + 16 * db + 4 * g4
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{T + 4 * q + j}, a{a + j}")
                e("s_nop 0")
                e(f"ds_write_b128 %[xout], v[{T + 4 * q}:{T + 4 * q + 3}] offset:{q * 1024}")
                q += 1
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e(f"ds_read_b128 v[{T + 68}:{T + 71}], %[xin] offset:8192")     # partner's (m', l')
        for q in range(8):
            e(f"ds_read_b128 v[{T + 4 * q}:{T + 4 * q + 3}], %[xin] offset:{q * 1024}")
        e("s_waitcnt lgkmcnt(8)")
        m, l, m2, l2 = T + 64, T + 65, T + 68, T + 69
        M, fa, fb = T + 72, T + 73, T + 74
        e(f"v_max_f32 v{M}, v{m}, v{m2}")
        e(f"v_sub_f32 v{fa}, v{m}, v{M}")
        e(f"v_sub_f32 v{fb}, v{m2}, v{M}")
        e(f"v_exp_f32 v{fa}, v{fa}")
        e(f"v_exp_f32 v{fb}, v{fb}")
        e("s_nop 1")
        e(f"v_mul_f32 v{l}, v{l}, v{fa}")
        e(f"v_fma_f32 v{l}, v{l2}, v{fb}, v{l}")
        e("s_nop 0")
        e(f"v_rcp_f32 v{l}, v{l}")
        e("s_nop 1")
        e(f"v_mul_f32 v{fa}, v{fa}, v{l}")      # own partial * 2^(m - M) / l
        e(f"v_mul_f32 v{fb}, v{fb}, v{l}")
        e("s_waitcnt lgkmcnt(0)")
        q = 0
        for db in keep:
            for g4 in range(4):
                t = T + 32 + 4 * (q % 4)
                pk = T + 76 + 2 * q
                a = O_SH + 16 * db + 4 * g4
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{t + j}, a{a + j}")
                e("s_nop 0")
                for j in range(4):
                    e(f"v_mul_f32 v{t + j}, v{t + j}, v{fa}")
                for j in range(4):
                    e(f"v_fma_f32 v{t + j}, v{T + 4 * q + j}, v{fb}, v{t + j}")
                e(f"v_cvt_pk_bf16_f32 v{pk}, v{t}, v{t + 1}")
                e(f"v_cvt_pk_bf16_f32 v{pk + 1}, v{t + 2}, v{t + 3}")
                e(f"buffer_store_dwordx2 v[{pk}:{pk + 1}], %[oos], s[60:63], 0 offen offset:{db * 64 + g4 * 16}")
                q += 1


# ---------------------------------------------------------------------------------------------------------------------
# static proof of the LDS ring's ordering
# ---------------------------------------------------------------------------------------------------------------------
class WaitCoverageError(AssertionError):
    pass


def check_wait_coverage(lines, iterations=3):
    """Every ds_read of a ring region (slot, K | V) must come behind a vmcnt wait that retired ALL LDS-DMA fills ever issued into
    that region and a barrier behind that wait (RAW); every LDS-DMA into a region behind an lgkmcnt wait that retired all earlier
    reads of it and a barrier behind that wait (WAR). One wave's program order stands for all four (same stream). The walk is
    prologue + `iterations` x loop body + epilogue with branches not taken (rare path and exits carry no ring traffic)."""
    try:
        i10 = lines.index("10:")
        ibr = lines.index("s_branch 10b")
    except ValueError as ex:
        raise WaitCoverageError("loop labels not found") from ex
    seq = lines[:i10] + lines[i10 + 1:ibr] * iterations + lines[ibr + 1:]
    vm, lg = [], []
    fills, reads = {}, {}
    m0 = None
    n_reads = n_dma = 0
    k_regs = {}
    for b, (hi, _h) in ((KA_LO, (0, 0)), (KB_LO, (0, 1)), (KA_HI, (1, 0)), (KB_HI, (1, 1))):
        for ks in range(8):
            k_regs[b + ks] = hi
    v_regs = {VA_LO + i: 0 for i in range(4)}
    v_regs.update({VA_HI + i: 1 for i in range(4)})
    in_exchange = False
    for pos, ins in enumerate(seq):
        mm = re.match(r"s_add_u32 m0, %\[wlds\], (\d+)", ins)
        if mm:
            m0 = int(mm.group(1))
            continue
        if ins.startswith("buffer_load_dwordx4") and ins.endswith("lds"):
            if m0 is None:
                raise WaitCoverageError(f"LDS-DMA without an m0 destination at {pos}: {ins}")
            region = (m0 // STAGE, "K" if (m0 % STAGE) < V_OFF else "V")
            for r in reads.get(region, []):
                if r["state"] != "fenced":
                    raise WaitCoverageError(f"WAR: LDS-DMA into {region} at {pos} ({ins}) while a ds_read of it issued at {r['pos']} "
                                            f"is only '{r['state']}' (needs lgkmcnt wait + barrier before the fill)")
            reads[region] = []
            n_dma += 1
            op = {"region": region, "state": "inflight", "pos": pos}
            vm.append(op)
            fills.setdefault(region, []).append(op)
            m0 = None
            continue
        if re.match(r"(global_load|buffer_load|global_store|buffer_store)", ins):
            vm.append({"region": None, "state": "inflight", "pos": pos})
            continue
        if ins.startswith("ds_write"):
            # exchange area (epilogue): legal only once every fill is visible and every ring read fenced
            for reg, ops in fills.items():
                for f in ops:
                    if f["state"] != "visible":
                        raise WaitCoverageError(f"exchange write at {pos} while the LDS-DMA of {reg} issued at {f['pos']} is '{f['state']}'")
            for reg, ops in reads.items():
                for r in ops:
                    if r["state"] != "fenced":
                        raise WaitCoverageError(f"exchange write at {pos} while a ring read of {reg} issued at {r['pos']} is '{r['state']}'")
            in_exchange = True
            lg.append({"region": None, "state": "issued", "pos": pos})
            continue
        mm = re.match(r"ds_read_b128 [av]\[\d+:\d+\], (\S+) offset:(\d+)", ins)
        if mm:
            addr, off = mm.group(1), int(mm.group(2))
            region = None
            hv = re.match(r"v(\d+)$", addr)
            if hv and int(hv.group(1)) in k_regs:
                region = (2 * k_regs[int(hv.group(1))] + off // STAGE, "K")
            elif hv and int(hv.group(1)) in v_regs:
                region = (2 * v_regs[int(hv.group(1))] + off // STAGE, "V")
            elif addr == "%[xin]":
                if not in_exchange:
                    raise WaitCoverageError(f"exchange read at {pos} before any exchange write")
            else:
                raise WaitCoverageError(f"unclassified LDS read at {pos}: {ins}")
            op = {"region": region, "state": "issued", "pos": pos}
            lg.append(op)
            if region is not None:
                n_reads += 1
                if not fills.get(region):
                    raise WaitCoverageError(f"RAW: ds_read of {region} at {pos} ({ins}) before anything was staged into it")
                for f in fills[region]:
                    if f["state"] != "visible":
                        raise WaitCoverageError(f"RAW: ds_read of {region} at {pos} ({ins}) while the LDS-DMA issued at {f['pos']} is "
                                                f"only '{f['state']}' (needs a covering vmcnt wait AND a barrier before the read)")
                reads.setdefault(region, []).append(op)
            continue
        if ins.startswith("s_waitcnt"):
            mv = re.search(r"vmcnt\((\d+)\)", ins)
            ml = re.search(r"lgkmcnt\((\d+)\)", ins)
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
            for ops in fills.values():
                for op in ops:
                    if op["state"] == "retired":
                        op["state"] = "visible"
            for ops in reads.values():
                for op in ops:
                    if op["state"] == "done":
                        op["state"] = "fenced"
            continue
    if n_reads == 0 or n_dma == 0:
        raise WaitCoverageError("checker saw no ring traffic - the stream format changed")
    return {"ring_reads": n_reads, "ring_fills": n_dma, "instructions": len(seq)}


def main():
    stamps = "--stamps" in sys.argv
    g = Gen(stamps)
    lines = g.build()
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc")
    name = "attention_x32_asm" + ("_stamps" if stamps else "") + ".inc"
    if stamps:
        for i in range(10):
            lines.insert(0, f"s_mov_b32 s{84 + i}, 0")
    lines = ["s_mov_b32 s96, m0"] + lines + ["s_mov_b32 m0, s96"]   # m0 saved / restored instead of clobbered (see gen_attn_w48.py)
    out = os.path.join(csrc, name)
    if "--inject-raw-race" in sys.argv:
        # leave the LDS-DMA of the step in flight across its barrier: the next step but one reads a slot that may be empty
        k = [i for i, ln in enumerate(lines) if ln == "s_waitcnt vmcnt(0)" and i > lines.index("10:")][0]
        lines[k] = f"s_waitcnt vmcnt({STAGE_OPS})"
    if "--inject-war-race" in sys.argv:
        # drop a step's barrier: the next step's LDS-DMA overwrites a slot whose reads were never fenced
        k = [i for i, ln in enumerate(lines) if ln == "s_barrier" and i > lines.index("10:")][0]
        lines[k] = "s_nop 0"
    stats = None
    if not (ABL - {""}):
        stats = check_wait_coverage(lines)
    header = "// GENERATED by tools/gen_attn_x32.py - do not edit. gfx950 assembly body of attn_fwd_kernel_x32_asm (attention.hip).\n"
    body = [header] + ['"' + ln.replace('"', '\\"') + '\\n\\t"\n' for ln in lines]
    if "--check" in sys.argv:
        same = os.path.exists(out) and open(out).read() == "".join(body)
        print(f"wait coverage ok: {stats}; committed file {'matches' if same else 'DIFFERS'}")
        sys.exit(0 if same or stamps else 4)
    with open(out, "w") as f:
        f.write("".join(body))
    if stamps:
        print(f"{len(lines)} lines -> {os.path.normpath(out)} (stamps build)")
        return
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(NA)] + [f"s{i}" for i in range(36, 97)] + ["vcc", "scc", "memory"]
    with open(os.path.join(csrc, "attention_x32_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_x32.py - do not edit. Registers the assembly body assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    n_mfma = sum(1 for ln in lines if "v_mfma" in ln)
    print(f"{len(lines)} lines, {n_mfma} MFMAs, {stats} -> {os.path.normpath(out)}")


if __name__ == "__main__":
    main()
