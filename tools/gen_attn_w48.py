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
  v[168:183]  (free; the fragment ring moved)    s[48:53]  compare masks of the reference check
  a[108:139]  K / Vt fragment ring (8 x 4): ds_read_b128 lands in AGPRs, the MFMAs read them as the A operand
  v[184:186]  per-lane maxima, v[187:194] rare-path temporaries, v195 floor, v[196:199] bf16 ones
  v[200:211]  -(reference max) per query block (x4: accumulator init of K.Q^T)        a[96:107] row sums l[qb]
  v[212:217]  fragment addresses + 64 KB (ring slots 2, 3)                              s[60:63] O buffer descriptor
Operands (compiler-assigned): see the asm statement in attention.hip.
"""
import os

QF, SA, SB, PF, RING = 0, 48, 96, 144, 168
MAXR = 184        # v184..186: this lane's maximum of each query block's scores (phase 1)
RT = 187          # v187..194: temporaries of the rare path and the prologue / epilogue
FLOOR = 195       # lower bound of the reference raise: -inf until the first tile set the reference, 0 afterwards
ONES = 196        # v[196:199]: bf16 ones, the A operand that makes the PV product also produce the row sums
NMCT = (200, 204, 208)  # per query block: -(reference maximum) in all four registers = accumulator init of the QK product
HI = 212          # v[212:217] fragment addresses + 64 KB (ring slots 2, 3)
NV = 220          # v0..v219 are assigned here; the compiler places the operands above
LACC = 96         # a[96:107]: row sums l[qb] (every register of a tuple holds the same value)
TAU = 8.0         # raise the reference only when a score exceeds it by more than 2^TAU: P stays <= 256
BIASR = 184       # masked variant: v[184:199] = this lane's 16 bias values of the tile being prepared (live from the start of part B1
                  # to the last bias add; shares v184..194 with MAXR / RT, which are only used afterwards)
BIAS_LDS = 4 * 32768  # byte offset of the bias vector in LDS (after the four K / Vt ring slots), Tk <= 4096 floats
BADDR = 221           # masked variant: LDS address of this lane's bias values of the tile being prepared
STAGE = 32768
if "--bias" in __import__("sys").argv:
    FLOOR, ONES, NV = 220, 224, 228   # v195..199 belong to the bias values in this variant
TM = 168             # v168, v169: temporaries of the ragged-tail mask (the old fragment ring's registers; the masked variant keeps
                     # its bias values in v[184:199] from the start of a step, so the mask must not use the rare-path temporaries)
ARING = LACC + 12   # a[108:139]: ring of RN K / Vt fragment registers (LDS reads land in AGPRs, MFMAs take them as their A operand)
RN = 8
RA = RN - 1  # fragment reads in flight ahead of their MFMAs. Stamps of the 4-register VGPR ring (RA = 3) showed every K.Q^T fragment
             # costing ~73 cycles against 48 of MFMA issue: with four waves reading and the LDS-DMA writing, an LDS read takes ~220
             # cycles to return, so the loop ran at RA reads per latency. Seven in flight cover it; the arch VGPRs have no room for
             # a deeper ring (v0..v219 + ~36 operands), the accumulator file has 148 registers to spare.
# Read pipeline across the step boundary (round 3, the 32x32 experiment's finding carried over): the first RA fragment reads of a step
# used to be issued behind the step's opening barrier and waited for at once - one exposed LDS latency (~200 cycles of ~2300) per tile
# step. Now a step's LDS-DMA is issued EARLY in the step (part A) and the step ends with vmcnt(0) + barrier, so at every barrier the
# three tiles ahead are visible and the first RA K reads of step t+1 can be issued under the last MFMAs of step t. W48_PIPE=step
# regenerates the old stream for A/B stamps.
CROSS = os.environ.get("W48_PIPE") != "step"
STAGE_OPS = 8  # LDS-DMA instructions per staged K / Vt tile (4 x 4 KB of K + 4 x 4 KB of Vt); Gen.stage() asserts it
TILES_AHEAD = 0 if CROSS else 1  # fills that may still be in flight when a step ends (W48_PIPE=step): the tile staged in this step (t+3) is first read two steps
                 # later, the one staged a step earlier (t+2) is read by the next step's K.Q^T and must have landed


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


BIAS = "--bias" in __import__("sys").argv   # masked variant: additive per-key bias (cross-attention text mask)
# prescaled variant: Q arrives multiplied by scale * log2(e) (rounded to bf16 once, by the producer): the MFMA's scores are already
# the base-2 exponents, so the 48 v_mul per tile (and the multiply of the rare path) disappear. %[c] is unused, %[tau] = 8.0,
# %[isc] = log2(e) (the bias is given in the reference's post-scale natural-log units).
PS = "--prescaled" in __import__("sys").argv
ABL = set(os.environ.get("W48_ABLATE", "").split(","))  # timing experiments only (wrong results): nodma, noexp, nords, ra6


class Gen:
    def __init__(self, stamps=False):
        self.lines = []
        self.stamps = stamps

    def stamp(self, i):
        """--stamps build only: s_memtime into s[54+2i : 55+2i] (written out at the end of the kernel)."""
        if self.stamps:
            self.e(f"s_memtime s[{64 + 2 * i}:{65 + 2 * i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    def e(self, s):
        self.lines.append(s)

    # ---- fragment reads: slot 0/1 use the operand addresses, slot 2/3 the +64 KB copies ----
    @staticmethod
    def ring(i):
        b = ARING + 4 * (i % RN)
        return f"a[{b}:{b + 3}]"

    def k_read(self, ring, slot, kb, ks):
        addr = f"%[ka{ks}]" if slot < 2 else f"v{HI + ks}"
        return f"ds_read_b128 {self.ring(ring)}, {addr} offset:{(slot & 1) * STAGE + kblock_off(kb)}"

    def v_read(self, ring, slot, db, i):
        addr = f"%[va{i}]" if slot < 2 else f"v{HI + 4 + i}"
        return f"ds_read_b128 {self.ring(ring)}, {addr} offset:{(slot & 1) * STAGE + vblock_off(db)}"

    def step_stream(self, sbuf, kslot, vslot, young=0):
        """One step's MFMA work as ONE read-ahead pipeline: S(next) = K Q^T from ring slot `kslot` (16 fragments), then O += Vt P from
        `vslot` (16 fragments). Fragment F lands in ring entry F % RN, its read is issued RA fragments ahead, so the Vt reads start
        under the last K.Q^T products and the LDS latency is exposed once per step (the first RA reads), not once per stream."""
        # k-step outer, key block inner: the four k-steps of one key block accumulate into the SAME three S tuples, and with only
        # three MFMAs between two links of that chain every fragment waited for the previous result (stamps: 75 cycles per fragment
        # against 48 of issue, 1230 cycles for the 48 products). Twelve MFMAs apart the chain never stalls.
        kf = [("k", kb, ks) for ks in range(4) for kb in range(4)]
        vf = [("v", db, i) for i in range(2) for db in range(8)]   # k-step 0 of every d-block first: it only needs pf[.][0]
        frags = kf + vf

        def read(F):
            if F >= len(frags):   # CROSS: the next step's first K fragments (K slot kslot + 1), same ring positions (32 % RN == 0)
                kind, x, y = frags[F - len(frags)]
                return ("ds", self.k_read(F, (kslot + 1) & 3, x, y))
            kind, x, y = frags[F]
            return ("ds", self.k_read(F, kslot, x, y) if kind == "k" else self.v_read(F, vslot, x, y))

        out = [] if CROSS else [read(F) for F in range(RA)]
        issued = RA
        for F, (kind, x, y) in enumerate(frags):
            # `young`: LDS reads issued behind the RA fragment reads that are already in flight at the step's start (the masked variant's
            # four bias reads): they are younger than fragments 0..RA-1 and older than everything issued in this step
            keep = issued - F - 1 + (young if F < RA else 0)
            out.append(("wait", f"s_waitcnt lgkmcnt({keep})"))
            for qb in range(3):
                if kind == "k":
                    d = vr(s_reg(sbuf, x, qb))
                    c = vr(NMCT[qb]) if y == 0 else d   # scores arrive as q.k - m_ref: the accumulators start at -m_ref
                    out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {d}, {self.ring(F)}, {vr(QF + (qb * 4 + y) * 4)}, {c}"))
                else:
                    a = f"a[{o_reg(x, qb)}:{o_reg(x, qb) + 3}]"
                    out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {a}, {self.ring(F)}, {vr(PF + (qb * 2 + y) * 4)}, {a}"))
            if F + RA < len(frags) or CROSS:
                out.append(read(F + RA))
                issued += 1
            if kind == "v" and x == 7:  # end of a k-step: the same P against a block of ones = this k-step's row sums
                for qb in range(3):
                    a = f"a[{LACC + 4 * qb}:{LACC + 4 * qb + 3}]"
                    out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {a}, {vr(ONES)}, {vr(PF + (qb * 2 + y) * 4)}, {a}"))
        return out

    def first_reads(self, kslot):
        """CROSS: the RA K reads a step expects in flight at its start (issued by the prologue for the first step)."""
        kf = [(kb, ks) for ks in range(4) for kb in range(4)]
        return [self.k_read(F, kslot, *kf[F]) for F in range(RA)]

    # ---- MFMA streams: list of groups, each group = [pre-instructions..., 3 MFMAs] per fragment ----
    def qk_stream(self, sbuf, slot):
        """S(next) = K Q^T from ring slot `slot` into S buffer `sbuf`. Returns a list of (kind, text); the first RA entries are
        the fragment reads issued ahead."""
        out = []
        frags = [(kb, ks) for ks in range(4) for kb in range(4)]   # k-step outer: see step_stream
        issued = 0
        for f in range(min(RA, 16)):
            out.append(("ds", self.k_read(f, slot, *frags[f])))
            issued += 1
        for f, (kb, ks) in enumerate(frags):
            out.append(("wait", f"s_waitcnt lgkmcnt({issued - f - 1})"))
            for qb in range(3):
                d = vr(s_reg(sbuf, kb, qb))
                c = vr(NMCT[qb]) if ks == 0 else d   # scores arrive as q.k - m_ref: the accumulators start at -m_ref
                out.append(("mfma", f"v_mfma_f32_16x16x32_bf16 {d}, {self.ring(f)}, {vr(QF + (qb * 4 + ks) * 4)}, {c}"))
            if f + RA < 16:
                out.append(("ds", self.k_read(f + RA, slot, *frags[f + RA])))
                issued += 1
        return out

    # ---- softmax of one S buffer (in place) ----
    # Scores arrive from the QK MFMAs already as s' = q.k - m_ref (raw score units relative to a REFERENCE maximum per query: the
    # accumulators start at -m_ref; the products stay exact, the scale c = scale*log2(e) is one v_mul before the exp). Pre-scaling Q
    # by c would save those 48 multiplies per tile but rounds q*c to bf16 a second time: 2-4 % error in P once |s*c| reaches ~30
    # (seen in test_attention_reference_maximum_stress) - not taken. The O rescale costs ~300 instructions for a wave's 96
    # accumulator registers (AGPR -> VGPR -> multiply -> AGPR), and with 48 queries per wave SOME query's maximum moves in nearly
    # every tile; so the reference is only raised when a score exceeds it by more than 2^TAU (after the scale): P = exp2(s'*c)
    # <= 2^TAU, l (a row of ones in the PV product) accumulates against the same reference, and the common path is 24 v_max3 + 3
    # compares, 48 v_mul, 48 v_exp and 24 v_cvt_pk per tile - no row-sum adds, no cross-lane reduction, no rescale.
    def bias_reads(self, slot_unused=None):
        """Masked variant: this lane's 16 bias values of the tile whose scores are about to be checked: keys 32(kb>>1) + 4(kb&1) +
        8g + j -> four ds_read_b128 at %[ba] (+ tile offset in s47) + 128 (kb>>1) + 16 (kb&1). They are the OLDEST LDS operations in
        flight wherever they are issued (before the fragment reads of the stream that follows), so the fragment waits cover them."""
        out = []
        for kb in range(4):
            out.append(f"ds_read_b128 v[{BIASR + 4 * kb}:{BIASR + 4 * kb + 3}], v{BADDR} offset:{128 * (kb >> 1) + 16 * (kb & 1)}")
        return out

    def bias_add(self, sbuf):
        """S' += bias / scale (the bias is given in post-scale units) for the three query blocks: the reference maximum, the threshold test and the exponentials all see
        the biased scores, so a fully masked row (-10000 on every key) still normalises like the reference's softmax."""
        out = []
        for kb in range(4):
            for j in range(4):
                for qb in range(3):
                    r = s_reg(sbuf, kb, qb, j)
                    out.append(f"v_fma_f32 v{r}, v{BIASR + 4 * kb + j}, %[isc], v{r}")   # + bias / scale
        return out

    def sm_max(self, sbuf):
        """Phase 1: per-lane maximum of each query block's 16 scores and the wave-wide 'exceeds threshold' masks."""
        lists = []
        for qb in range(3):
            t = MAXR + qb
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
            out.append(f"v_cmp_gt_f32_e64 s[{48 + 2 * qb}:{49 + 2 * qb}], v{t}, %[tau]")   # 8.0 is not an inline constant
            lists.append(out)
        return [x[k] for k in range(len(lists[0])) for x in lists]

    def sm_exp(self, sbuf, half):
        """Phase 2 for key blocks 2*half, 2*half+1 (= k-step `half` of the PV product): P = exp2(s' * c), bf16 pack into pf[qb][half]."""
        lists = []
        for qb in range(3):
            tup = [s_reg(sbuf, 2 * half + hh, qb) for hh in range(2)]
            out = []
            if not PS:
                for b in tup:
                    for j in range(4):
                        out.append(f"v_mul_f32 v{b + j}, %[c], v{b + j}")
            for b in tup:
                for j in range(4):
                    out.append(f"v_exp_f32 v{b + j}, v{b + j}")
            p = PF + (qb * 2 + half) * 4
            for hh, b in enumerate(tup):
                out.append(f"v_cvt_pk_bf16_f32 v{p + 2 * hh}, v{b}, v{b + 1}")
                out.append(f"v_cvt_pk_bf16_f32 v{p + 2 * hh + 1}, v{b + 2}, v{b + 3}")
            lists.append(out)
        return [x[k] for k in range(len(lists[0])) for x in lists]

    def tail_mask(self, label, sbuf, count):
        """If exactly `count` tiles remain after... (s46 == count) and the key count is ragged: the tile in `sbuf` is the last one -
        its keys at or beyond Tk (zero rows from the K descriptor) get a score of -inf. %[tmask] bit kb*4+j = that key is invalid."""
        e = self.e
        e(f"s_cmp_eq_u32 s46, {count}")
        e(f"s_cbranch_scc0 {label}f")
        e("s_cmp_eq_u32 %[rag], 0")
        e(f"s_cbranch_scc1 {label}f")
        e("s_nop 7")
        e("s_nop 7")
        e(f"v_mov_b32 v{TM + 1}, 0xff800000")
        for kb in range(4):
            for j in range(4):
                e(f"v_and_b32 v{TM}, {1 << (kb * 4 + j)}, %[tmask]")
                e(f"v_cmp_ne_u32_e64 s[48:49], 0, v{TM}")
                e("s_nop 1")
                for qb in range(3):
                    r = s_reg(sbuf, kb, qb, j)
                    e(f"v_cndmask_b32_e64 v{r}, v{r}, v{TM + 1}, s[48:49]")
        e(f"{label}:")

    def sm_check_and_rare_path(self, label, sbuf, force=False):
        """After phase 1 on `sbuf`: if any lane saw a score above the threshold, raise every query's reference to its running
        maximum: delta = max(row maximum, floor) in shifted units; S' -= delta, accumulator init -= delta, O and l *= 2^-delta."""
        e = self.e
        e("s_nop 3")
        e("s_or_b64 s[48:49], s[48:49], s[50:51]")
        e("s_or_b64 s[48:49], s[48:49], s[52:53]")
        e("s_cmp_lg_u64 s[48:49], 0")
        if not force:  # the first tile always sets the reference to the true row maximum (the floor is -inf then): without it a row
            e(f"s_cbranch_scc0 {label}f")  # whose scores are all far below zero (a fully masked row: -10000) would underflow to l = 0
        for qb in range(3):
            t = MAXR + qb
            for swap in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):  # maximum over the four lanes of a query
                e(f"v_mov_b32 v{RT}, v{t}")
                e("s_nop 1")
                e(f"{swap} v{t}, v{RT}")
                e("s_nop 1")
                e(f"v_max_f32 v{t}, v{t}, v{RT}")
            e(f"v_max_f32 v{RT + 1}, v{t}, v{FLOOR}")             # delta
            if PS:
                e(f"v_sub_f32 v{RT + 2}, 0, v{RT + 1}")            # scores are base-2 exponents already
            else:
                e(f"v_mul_f32 v{RT + 2}, %[c], v{RT + 1}")
                e(f"v_sub_f32 v{RT + 2}, 0, v{RT + 2}")
            e(f"v_min_f32 v{RT + 2}, 0, v{RT + 2}")                # first tile: the reference may move DOWN (O = l = 0 then; a
            e(f"v_exp_f32 v{RT + 2}, v{RT + 2}")                   # fully masked row would give 2^14427 = inf, inf * 0 = NaN)
            for j in range(4):
                e(f"v_sub_f32 v{NMCT[qb] + j}, v{NMCT[qb] + j}, v{RT + 1}")
            for kb in range(4):
                for j in range(4):
                    r = s_reg(sbuf, kb, qb, j)
                    e(f"v_sub_f32 v{r}, v{r}, v{RT + 1}")
            if force:
                continue  # first tile: O and l are still zero, nothing to rescale
            e("s_nop 7")
            e("s_nop 7")
            for a in [o_reg(db, qb) for db in range(8)] + [LACC + 4 * qb]:
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{RT + 4 + j}, a{a + j}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_mul_f32 v{RT + 4 + j}, v{RT + 4 + j}, v{RT + 2}")
                e("s_nop 1")
                for j in range(4):
                    e(f"v_accvgpr_write_b32 a{a + j}, v{RT + 4 + j}")
        e(f"v_mov_b32 v{FLOOR}, 0")
        e("s_nop 7")
        e(f"{label}:")

    def interleave(self, stream, valu, per_mfma):
        """Emit (part of) an MFMA stream with `per_mfma` instructions of `valu` after every MFMA; leftovers at the end."""
        vi = 0
        for kind, text in stream:
            if "nords" in ABL and kind in ("ds", "wait"):
                continue
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
    def issue_cost(group):
        """Vector-issue cycles of a filler group beside MFMAs (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
        c = 0
        for ins in group:
            if ins.startswith(("v_exp_f32", "v_rcp_f32", "v_log_f32")):
                c += 8
            elif ins.startswith("buffer_load"):
                c += int(os.environ.get("W48_DMA_COST", "24"))
            elif ins.startswith(("s_nop", "s_add", "s_cmp", "s_sub")):
                c += 1
            else:
                c += 4
        return c

    def spread_gaps(self, stream, groups, over):
        """Cost-aware placement: an MFMA holds the vector issue port for 8 of its 16 cycles, so a gap BETWEEN TWO MFMAs hides 8 cycles
        of fillers; a gap that already carries a fragment read / wait hides none. Fillers go, in order, into the pure gaps of the first
        `over` MFMAs in proportion to their issue cost; busy gaps only take what the pure ones cannot."""
        items = [(k, t) for k, t in stream if not ("nords" in ABL and k in ("ds", "wait"))]
        mf = [i for i, (k, _) in enumerate(items) if k == "mfma"]
        pure = [i for n, i in enumerate(mf[:over]) if i + 1 < len(items) and items[i + 1][0] == "mfma"]
        busy = [i for n, i in enumerate(mf[:over]) if not (i + 1 < len(items) and items[i + 1][0] == "mfma")]
        bcap = float(os.environ.get("W48_BUSY_CAP", "0"))   # cycles a gap that carries a fragment read / wait may still take
        demand = sum(self.issue_cost(g) for g in groups)
        cap = 8.0 * len(pure) + bcap * len(busy)
        f = max(1.0, demand / max(cap, 1.0))
        gi, cum, allowed = 0, 0.0, 0.0
        pure_set, busy_set = set(pure), set(busy)
        for i, (kind, text) in enumerate(items):
            self.e(text)
            if i in pure_set or (bcap > 0 and i in busy_set):
                allowed += (8.0 if i in pure_set else bcap) * f
                while gi < len(groups) and cum + self.issue_cost(groups[gi]) <= allowed + 0.5:
                    if i in busy_set and self.issue_cost(groups[gi]) > 4:
                        break  # no 8-cycle instruction beside a fragment read
                    for ins in groups[gi]:
                        self.e(ins)
                    cum += self.issue_cost(groups[gi])
                    gi += 1
        while gi < len(groups):
            for ins in groups[gi]:
                self.e(ins)
            gi += 1

    def spread(self, stream, groups, over):
        """Emit an MFMA stream with the instruction groups spread evenly behind its first `over` MFMAs (leftovers at the end)."""
        if os.environ.get("W48_SPREAD") != "even":  # "even": the round-1 placement (one filler behind every MFMA), kept for A/B stamps
            return self.spread_gaps(stream, groups, over)
        gi, nm = 0, 0
        for kind, text in stream:
            if "nords" in ABL and kind in ("ds", "wait"):
                continue
            self.e(text)
            if kind == "mfma":
                nm += 1
                want = (len(groups) * nm + over - 1) // over if nm <= over else len(groups)
                while gi < min(want, len(groups)):
                    for ins in groups[gi]:
                        self.e(ins)
                    gi += 1
        while gi < len(groups):
            for ins in groups[gi]:
                self.e(ins)
            gi += 1

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
        assert sum(ins.startswith("buffer_load") for grp in out for ins in grp) == STAGE_OPS
        return out

    def advance_stage_offsets(self):
        self.e("s_add_u32 s44, s44, %[ktb]")
        self.e("s_add_u32 s45, s45, 128")

    def step(self, slot, cur, nxt, uid):
        self.e(f"; ---------------- tile step, ring slot {slot} ----------------")
        st = self.stamp if slot == 0 else (lambda i: None)
        st(0)
        # Entry state: S(t) in `cur`, already checked against the reference maxima (end of the previous step / prologue).
        # part A : S(t+1) = K Q^T (48 MFMAs) with P(t) = exp2(S(t)*c - m_ref*c), row sums and the bf16 pack in the gaps, two VALU
        #          instructions per MFMA (what a 4-pass MFMA hides, tools/ubench/valu_slots.hip)
        # part B1: O += Vt P, k-step 0 of every d-block (24 MFMAs) with the rest of that VALU work (key blocks 2, 3 -> pf[.][1])
        # part B2: k-step 1 (24 MFMAs) with the per-lane maxima of S(t+1) and the LDS-DMA of tile t+3 in the gaps
        # then the reference check for S(t+1) (rare path: rescale O and l), counted vmcnt, barrier
        va = [] if "noexp" in ABL else self.sm_exp(cur, 0)   # -> pf[.][0], needed by part B1
        vb = [] if "noexp" in ABL else self.sm_exp(cur, 1)   # -> pf[.][1], needed by part B2
        # balance: part A has 48 MFMAs, part B1 24 - the multiplies of the second half move up into part A
        mul_b = [x for x in vb if x.startswith("v_mul_f32")]
        vb = [x for x in vb if not x.startswith("v_mul_f32")]
        va = va + mul_b
        allst = self.step_stream(nxt, (slot + 1) & 3, slot, young=4 if (BIAS and CROSS) else 0)
        part_a, rest = self.split_stream(allst, 48)
        head, tail = self.split_stream(rest, 24)
        if BIAS:  # bias of tile t+1: its reads come first in the step (older than every fragment read issued in it, so the fragment
            self.e(f"v_add_u32 v{BADDR}, 256, v{BADDR}")   # waits cover them; v184..199 are free here: maxima / rare-path temporaries are dead)
            for ins in self.bias_reads():
                self.e(ins)
            vb = vb + self.bias_add(nxt)
        dma = [] if "nodma" in ABL else self.stage((slot + 3) & 3)
        if CROSS:
            # LDS-DMA of tile t+3 early in the step, spread through part A between the exponentials
            fill = [[x] for x in va]
            stride = max(1, len(fill) // (len(dma) + 1)) if dma else 1
            for k, grp in enumerate(dma):
                fill.insert(min(len(fill), (k + 1) * stride + k), grp)
            dma = []
            self.spread(part_a, fill, 46)
        else:
            for kind, text in part_a[:RA]:
                self.e(text)
            npre = min(6, len(va))
            for ins in va[:npre]:  # independent of the reads just issued: covers part of their latency
                self.e(ins)
            self.spread(part_a[RA:], [[x] for x in va[npre:]], 44)
        self.tail_mask(f"{uid + 10}", nxt, 2)   # S(t+1) is the last, ragged tile
        st(1)
        self.spread(head, [[x] for x in vb], 22)
        st(2)
        mx = self.sm_max(nxt)
        work = []  # B2 filler: maxima instructions (and, W48_PIPE=step, the DMA groups)
        while mx or dma:
            for _ in range(4):
                if mx:
                    work.append([mx.pop(0)])
            if dma:
                work.append(dma.pop(0))
        self.spread(tail, work, 26)
        # tile t was the last one: leave before the reference check of a tile that does not exist (its zero scores could raise the
        # reference of a row whose real scores are all far below zero, scaling O and l to nothing)
        self.e("s_sub_u32 s46, s46, 1")
        self.e("s_cmp_eq_u32 s46, 0")
        self.e("s_cbranch_scc1 30f")
        self.sm_check_and_rare_path(f"{uid}", nxt)
        self.advance_stage_offsets()
        st(3)
        # counted wait derived from the staging model (not a literal): everything but the youngest TILES_AHEAD fills has landed.
        # check_wait_coverage() proves that every ring-slot read of the next steps is covered by this wait + barrier.
        self.e(f"s_waitcnt vmcnt({STAGE_OPS * TILES_AHEAD})")
        self.e("s_barrier")
        st(4)

    def kstamp(self, i):
        """--stamps build: whole-kernel stamps (start, prologue done, loop done, stores issued) in s[76+2i : 77+2i]."""
        if self.stamps:
            self.e(f"s_memtime s[{76 + 2 * i}:{77 + 2 * i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    def build(self):
        e = self.e
        self.kstamp(0)
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
        e("s_mov_b32 s46, %[nt]")        # tiles left (including the one whose S is current)
        e("s_mov_b32 s60, %[oblo]")
        e("s_mov_b32 s61, %[obhi]")
        e("s_mov_b32 s62, %[orec]")      # rows at or beyond Tq fall outside the descriptor: their stores are dropped
        e("s_mov_b32 s63, 0x00020000")
        for ks in range(4):
            e(f"v_add_u32 v{HI + ks}, 0x10000, %[ka{ks}]")
        for i in range(2):
            e(f"v_add_u32 v{HI + 4 + i}, 0x10000, %[va{i}]")
        # the long-latency work first: the Q fragments (and the bias vector), then K / Vt tiles 0..2, then register init. ALL three
        # tiles must have landed before the loop: its first step already multiplies Q with the keys of tile 1 (part A works one tile
        # ahead of the PV products), so a wait that leaves tiles 1 and 2 in flight reads a ring slot that may still be empty.
        e("; ---- Q fragments ----")
        for qb in range(3):
            for ks in range(4):
                e(f"global_load_dwordx4 {vr(QF + (qb * 4 + ks) * 4)}, %[qo{qb}], %[qbase] offset:{ks * 64}")
        if BIAS:
            e("; ---- bias vector -> LDS (16 KB after the ring; the descriptor ends after key Tk-1: zeros beyond) ----")
            e("s_mov_b32 s56, %[bilo]")
            e("s_mov_b32 s57, %[bihi]")
            e("s_mov_b32 s58, %[birec]")
            e("s_mov_b32 s59, 0x00020000")
            for i in range(4):
                e(f"s_add_u32 m0, %[wlds], {BIAS_LDS + i * 4096}")
                e(f"s_add_u32 s47, %[wlds], {i * 4096}")
                e("buffer_load_dwordx4 %[bvo], s[56:59], s47 offen lds")
            e(f"v_mov_b32 v{BADDR}, %[ba]")
        # Every CU starts at once and the memory system serves the burst roughly first come, first served: with Q and three tiles
        # requested up front (144 KB per CU, 37 MB in all) the first K.Q^T waited for most of it. Only Q, K0 and V0 are requested
        # before the first wait; tiles 1 and 2 are staged in the gaps of the first K.Q^T product (W48_PROLOGUE=early restores the
        # old order for A/B stamps).
        late = os.environ.get("W48_PROLOGUE") != "early"
        e("; ---- tile 0" + ("" if late else ", 1, 2") + " ----")
        n_stage_ops = 0
        later = []
        for t in range(3):
            grps = self.stage(t)
            n_stage_ops += sum(ins.startswith("buffer_load") for grp in grps for ins in grp)
            adv = ["s_add_u32 s44, s44, %[ktb]", "s_add_u32 s45, s45, 128"]
            if late and t > 0:
                later += grps
                later[-1] = later[-1] + adv   # the offsets advance behind the tile's last piece
            else:
                for grp in grps:
                    for ins in grp:
                        e(ins)
                for ins in adv:
                    e(ins)
        assert n_stage_ops == 3 * STAGE_OPS, n_stage_ops
        for i in range(LACC + 12):
            e(f"v_accvgpr_write_b32 a{i}, 0")
        for qb in range(3):
            for j in range(4):
                e(f"v_mov_b32 v{NMCT[qb] + j}, 0")
        e(f"v_mov_b32 v{FLOOR}, 0xff800000")
        for j in range(4):
            e(f"v_mov_b32 v{ONES + j}, 0x3f803f80")
        # Counted waits derived from the issue order Q, [bias,] K0, V0, K1, V1, K2, V2 (in-order counter): the first K.Q^T product
        # needs Q and K0 only, so it runs while V0 / K1 are still landing; the loop's first step reads K of slot 1 (part A works one
        # tile ahead) and Vt of slot 0, so V1 and all of tile 2 may still be in flight when it starts - the counted wait + barrier at
        # the end of step 0 retires them before step 1 reads them. check_wait_coverage() proves both (a prologue that left K1 in
        # flight shipped once and raced: DESIGN.md).
        half = STAGE_OPS // 2
        e(f"s_waitcnt vmcnt({(STAGE_OPS if late else 3 * STAGE_OPS) - half})")     # everything up to K0 has landed
        e("s_barrier")
        if late:
            self.spread_gaps(self.qk_stream(SA, 0), later, 40)
        else:
            for kind, text in self.qk_stream(SA, 0):
                e(text)
        e("s_nop 7")
        e("s_nop 7")
        self.tail_mask("8", SA, 1)       # a single, ragged tile
        if BIAS:
            for ins in self.bias_reads():
                e(ins)
            e("s_waitcnt lgkmcnt(0)")
            for ins in self.bias_add(SA):
                e(ins)
        for ins in self.sm_max(SA):
            e(ins)
        self.sm_check_and_rare_path("9", SA, force=True)
        if CROSS:   # the first step issues the K reads of tile 2 before its closing barrier: K2 must be visible here; only V2 may fly
            e(f"s_waitcnt vmcnt({half})")
            e("s_barrier")
            for r in self.first_reads(1):
                e(r)
        else:
            e(f"s_waitcnt vmcnt({STAGE_OPS + half})")          # V0 and K1 have landed; V1, K2, V2 may fly
            e("s_barrier")
        self.kstamp(1)
        e("10:")
        self.step(0, SA, SB, 11)
        self.step(1, SB, SA, 12)
        self.step(2, SA, SB, 13)
        self.step(3, SB, SA, 14)
        e("s_branch 10b")
        e("30:")
        self.kstamp(2)
        e("; ---- epilogue: normalise, stage the wave's 48 output rows in LDS, store whole rows ----")
        # O leaves the accumulators as 8-byte pieces at a 8 KB row stride (a lane owns 4 consecutive d of one query): stored like that,
        # every buffer_store touches 16 rows x 32 B and the kernel's tail is store-issue bound (~9 k cycles). Instead each wave writes
        # its [48 rows][256 B] block to LDS (ring slots 0 / 1 are free behind the barrier; 16-byte chunks XOR-swizzled by the row so that
        # both the 8-byte writes and the 16-byte row reads are conflict-free) and stores 4 full rows per instruction (dwordx4).
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")   # LDS-DMA issued past the last tile, fragment reads issued ahead
        e("s_barrier")              # every wave has left the ring
        e("s_nop 7")
        e("s_nop 7")
        # lane geometry in registers of its own (no operands: the compiler has few VGPRs left beside the ~225 this stream owns)
        LN, C16, G, EST, ESX, ERB, ERX, ESO = (SB + 48 + k for k in range(8))   # v[144:151] (P fragments: dead)
        e(f"v_mbcnt_lo_u32_b32 v{LN}, -1, 0")
        e(f"v_mbcnt_hi_u32_b32 v{LN}, -1, v{LN}")
        e(f"v_and_b32 v{C16}, 15, v{LN}")
        e(f"v_lshrrev_b32 v{G}, 4, v{LN}")
        e("s_mul_i32 s47, %[wlds], 12")                       # wave * 12288: the wave's staging block
        # write side: lane (c16, g) holds 4 consecutive d of row 16 qb + c16 per (qb, db): 8 bytes at chunk (2 db + g / 2) ^ c16
        #   est = wave * 12288 + c16 * 256 + (g & 1) * 8 ; esx = ((g >> 1) ^ c16) << 4 ; address = est + qb * 4096 + ((db << 5) ^ esx)
        e(f"v_lshlrev_b32 v{EST}, 8, v{C16}")
        e(f"v_and_b32 v{RT}, 1, v{G}")
        e(f"v_lshl_add_u32 v{EST}, v{RT}, 3, v{EST}")
        e(f"v_add_u32 v{EST}, s47, v{EST}")
        e(f"v_lshrrev_b32 v{ESX}, 1, v{G}")
        e(f"v_xor_b32 v{ESX}, v{ESX}, v{C16}")
        e(f"v_lshlrev_b32 v{ESX}, 4, v{ESX}")
        # read side: lane (lr = g, c = c16) takes chunk c of rows 4 i + lr: erb + (i & 3) * 1024 + (erx ^ ((i & 3) << 6)) + (i >> 2) * 4096
        #   erb = wave * 12288 + lr * 256 ; erx = (c ^ lr) << 4
        e(f"v_lshlrev_b32 v{ERB}, 8, v{G}")
        e(f"v_add_u32 v{ERB}, s47, v{ERB}")
        e(f"v_xor_b32 v{ERX}, v{C16}, v{G}")
        e(f"v_lshlrev_b32 v{ERX}, 4, v{ERX}")
        # global row offset of the lane's first row group: (q0w + lr) * ldo * 2 + c * 16, q0w * ldo * 2 = %[eso] (scalar), ldo * 2 = %[o1]
        e(f"v_mul_lo_u32 v{ESO}, v{G}, %[o1]")
        e(f"v_lshl_add_u32 v{ESO}, v{C16}, 4, v{ESO}")
        e(f"v_add_u32 v{ESO}, %[eso], v{ESO}")
        # key-split launches (%[lse] != 0: few queries, the keys divided over several workgroups): besides its normalised O, which then
        # goes to the split's slice of a partial buffer, the wave stores log2 of each query's denominator in absolute units,
        # log2(l) + m_ref (* c when the scores are not base-2 exponents yet): the weights of attn_combine_kernel. The four lanes that
        # share a query store the same value to the same address.
        e("s_cmp_eq_u64 %[lse], 0")
        e("s_cbranch_scc1 31f")
        e(f"v_lshlrev_b32 v{RT + 1}, 2, v{C16}")
        for qb in range(3):
            d = RT + 2 + qb          # a data register per store: nothing rewrites it while the store is in flight
            e(f"v_accvgpr_read_b32 v{d}, a{LACC + 4 * qb}")
            e("s_nop 1")
            e(f"v_log_f32 v{d}, v{d}")
            if PS:
                e("s_nop 1")
                e(f"v_sub_f32 v{d}, v{d}, v{NMCT[qb]}")
            else:
                e(f"v_mul_f32 v{RT + 6}, %[c], v{NMCT[qb]}")
                e(f"v_sub_f32 v{d}, v{d}, v{RT + 6}")
            e(f"global_store_dword v{RT + 1}, v{d}, %[lse] offset:{qb * 64}")
        e("31:")
        WA = SA                      # v[48:55]: write address per db
        for db in range(8):
            e(f"v_xor_b32 v{WA + db}, {db << 5}, v{ESX}")
            e(f"v_add_u32 v{WA + db}, v{WA + db}, v{EST}")
        T = SA + 8                   # temporaries v[56:...]
        for qb in range(3):
            e(f"v_accvgpr_read_b32 v{RT}, a{LACC + 4 * qb}")
            e("s_nop 1")
            e(f"v_rcp_f32 v{RT}, v{RT}")
            e("s_nop 1")
            for db in range(8):
                a = o_reg(db, qb)
                t = T + 6 * ((qb * 8 + db) % 6)
                for j in range(4):
                    e(f"v_accvgpr_read_b32 v{t + j}, a{a + j}")
                e("s_nop 0")
                for j in range(4):
                    e(f"v_mul_f32 v{t + j}, v{t + j}, v{RT}")
                e(f"v_cvt_pk_bf16_f32 v{t + 4}, v{t}, v{t + 1}")
                e(f"v_cvt_pk_bf16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
                e(f"ds_write_b64 v{WA + db}, v[{t + 4}:{t + 5}] offset:{qb * 4096}")
        e("s_waitcnt lgkmcnt(0)")    # a wave reads back only what it wrote itself: no barrier
        # read row group i, store 4 whole rows per instruction
        R = SB                       # v[96:143]: 12 x 4 registers
        RA4 = SA + 8                 # v[56:59]: read address per i & 3 (the write temporaries are dead)
        for n in range(4):
            e(f"v_xor_b32 v{RA4 + n}, {n << 6}, v{ERX}")
            e(f"v_add_u32 v{RA4 + n}, v{RA4 + n}, v{ERB}")
        for i in range(12):
            e(f"ds_read_b128 v[{R + 4 * i}:{R + 4 * i + 3}], v{RA4 + (i & 3)} offset:{(i >> 2) * 4096 + (i & 3) * 1024}")
        O0 = SA + 12                 # v60, v61: two offset registers (the store in flight keeps its own)
        e(f"v_mov_b32 v{O0}, v{ESO}")
        for i in range(12):
            cur, nxt = O0 + (i & 1), O0 + ((i + 1) & 1)
            if i < 11:
                e(f"v_add_u32 v{nxt}, %[o4], v{cur}")
            e(f"s_waitcnt lgkmcnt({11 - i})")
            e(f"buffer_store_dwordx4 v[{R + 4 * i}:{R + 4 * i + 3}], v{cur}, s[60:63], 0 offen")
        # no wait for the stores: they may complete after the wave ends (the LDS-DMA issued past the last tile was waited for above)
        if self.stamps:
            e("s_waitcnt vmcnt(0)")
        self.kstamp(3)
        if self.stamps:  # lane 0 of every wave writes its 5 stamps: dbg[wave][5] u64
            for i in range(10):
                e(f"v_mov_b32 v{PF + i}, s{64 + i}")
            e(f"v_mov_b32 v{PF + 10}, 0")
            for i in range(5):
                e(f"global_store_dwordx2 v{PF + 10}, v[{PF + 2 * i}:{PF + 2 * i + 1}], %[dbg] offset:{i * 8}")
            e("s_waitcnt vmcnt(0)")
            for i in range(8):
                e(f"v_mov_b32 v{PF + i}, s{76 + i}")
            for i in range(4):
                e(f"global_store_dwordx2 v{PF + 10}, v[{PF + 2 * i}:{PF + 2 * i + 1}], %[dbg] offset:{128 + i * 8}")
            e("s_waitcnt vmcnt(0)")
        return self.lines


class WaitCoverageError(AssertionError):
    pass


def check_wait_coverage(lines, iterations=3):
    """Static proof that the emitted stream orders every LDS-DMA fill against the ds_reads of its ring slot.

    The four waves of a workgroup run this same stream, so one wave's program order stands for all of them. Rules (MI355X guide,
    'Read a staged buffer one phase AFTER the wait that retires it'; LDS-DMA is ordered for a ds_read only by the issuing waves'
    counted vmcnt followed by a barrier the reader has passed):
      RAW  a ds_read of ring region (slot, K|Vt) needs every LDS-DMA ever issued into that region to have been retired by an
           `s_waitcnt vmcnt(n)` (in-order counter: all but the n youngest vector-memory operations are done) that is itself followed
           by an `s_barrier`, both before the read in program order.
      WAR  an LDS-DMA into a region needs every earlier ds_read of that region to have been retired by an `s_waitcnt lgkmcnt(n)`
           and then an `s_barrier` - before the DMA is issued.
    The stream is walked as prologue + `iterations` x loop body (+ epilogue), branches not taken: the rare path and the ragged-tail
    code contain no memory operations, and the exit branch only skips code. Raises WaitCoverageError naming the first violation."""
    import re

    try:
        i10 = lines.index("10:")
        ibr = lines.index("s_branch 10b")
    except ValueError as e:
        raise WaitCoverageError("loop labels not found") from e
    seq = lines[:i10] + lines[i10 + 1:ibr] * iterations + lines[ibr + 1:]
    vm = []      # outstanding vector-memory operations in issue order: dicts {region|None, state}
    lg = []      # outstanding LDS reads in issue order
    fills = {}   # region -> list of DMA ops (all generations)
    reads = {}   # region -> list of read ops
    m0 = None
    n_reads = n_dma = 0
    for pos, ins in enumerate(seq):
        mm = re.match(r"s_add_u32 m0, %\[wlds\], (\d+)", ins)
        if mm:
            m0 = int(mm.group(1))
            continue
        if ins.startswith("buffer_load_dwordx4") and ins.endswith("lds"):
            if m0 is None:
                raise WaitCoverageError(f"LDS-DMA without an m0 destination at {pos}: {ins}")
            region = None
            if m0 < 4 * STAGE:
                region = (m0 // STAGE, "K" if (m0 % STAGE) < 16384 else "V")
                for r in reads.get(region, []):
                    if r["state"] != "fenced":
                        raise WaitCoverageError(f"WAR: LDS-DMA into {region} at {pos} ({ins}) while a ds_read of it issued at "
                                                f"{r['pos']} is only '{r['state']}' (needs lgkmcnt wait + barrier before the fill)")
                reads[region] = []
                n_dma += 1
            elif m0 >= BIAS_LDS:
                region = ("bias", "vector")   # staged once in the prologue, read by every step of the masked variant
            op = {"region": region, "state": "inflight", "pos": pos}
            vm.append(op)
            if region is not None:
                fills.setdefault(region, []).append(op)
            m0 = None
            continue
        if re.match(r"(global_load|buffer_load|global_store|buffer_store)", ins):
            vm.append({"region": None, "state": "inflight", "pos": pos})
            continue
        mm = re.match(r"ds_read_b128 [av]\[\d+:\d+\], (\S+) offset:(\d+)", ins) or re.match(r"ds_read_b128 [av]\[\d+:\d+\], (\S+)$", ins)
        if mm:
            addr = mm.group(1)
            off = int(mm.group(2)) if mm.lastindex == 2 else 0
            region = None
            ka = re.match(r"%\[ka(\d)\]", addr)
            va = re.match(r"%\[va(\d)\]", addr)
            hv = re.match(r"v(\d+)$", addr)
            if ka:
                region = (off // STAGE, "K")
            elif va:
                region = (off // STAGE, "V")
            elif hv and HI <= int(hv.group(1)) < HI + 6:
                region = (2 + off // STAGE, "K" if int(hv.group(1)) < HI + 4 else "V")
            elif hv and int(hv.group(1)) == BADDR:
                region = ("bias", "vector")
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
            # generations that are fully visible and superseded need no further tracking, but keeping them is harmless
            continue
    if n_reads == 0 or n_dma == 0:
        raise WaitCoverageError("checker saw no ring traffic - the stream format changed")
    return {"ring_reads": n_reads, "ring_fills": n_dma, "instructions": len(seq)}


def main():
    import sys
    stamps = "--stamps" in sys.argv
    g = Gen(stamps)
    lines = g.build()
    # m0 (the LDS-DMA destination) is a reserved register: naming it in the clobber list draws "-Winline-asm ... may lead to undefined
    # behaviour" from the compiler (round-4 verdict, Weak 10). The block saves it in an SGPR of its own and restores it on the way out,
    # so the surrounding HIP code sees m0 unchanged and the clobber list no longer names it.
    lines = ["s_mov_b32 s95, m0"] + lines + ["s_mov_b32 m0, s95"]   # (s84..s93 hold the stamps of the --stamps build)
    here = os.path.dirname(os.path.abspath(__file__))
    name = "attention_w48_asm" + ("_bias" if BIAS else "") + ("_ps" if PS else "") + ("_stamps" if stamps else "") + ".inc"
    out = os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc", name)
    if "--inject-prologue-race" in sys.argv:
        # the race that shipped once (commit 7c76101): the prologue left tiles 1 and 2 in flight while step 0 reads slot 1
        want = f"s_waitcnt vmcnt({STAGE_OPS // 2})" if CROSS else f"s_waitcnt vmcnt({STAGE_OPS + STAGE_OPS // 2})"
        k = max(i for i, ln in enumerate(lines[:lines.index("10:")]) if ln == want)
        lines[k] = f"s_waitcnt vmcnt({2 * STAGE_OPS})"
    stats = None
    if not (ABL - {""}):
        stats = check_wait_coverage(lines)   # every generated stream is proven before it is written
    if "--check" in sys.argv:
        # CPU test entry: prove the stream and compare it with the committed file; writes nothing
        body = ["// GENERATED by tools/gen_attn_w48.py - do not edit. gfx950 assembly body of attn_fwd_kernel_w48_asm (attention.hip).\n"]
        body += ['"' + ln.replace('"', '\\"') + '\\n\\t"\n' for ln in lines]
        same = os.path.exists(out) and open(out).read() == "".join(body)
        print(f"wait coverage ok: {stats}; committed file {'matches' if same else 'DIFFERS'}")
        sys.exit(0 if same or stamps else 4)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_w48.py - do not edit. gfx950 assembly body of attn_fwd_kernel_w48_asm (attention.hip).\n")
        for ln in lines:
            f.write('"' + ln.replace('"', '\\"') + '\\n\\t"\n')
    clob = [f"v{i}" for i in range(NV)] + [f"a{i}" for i in range(ARING + 4 * RN)] + [f"s{i}" for i in range(36, 84)] + ["s95", "vcc", "scc", "memory"]
    with open(os.path.join(here, "..", "ltx-video-swift-mlx_amd", "csrc", "attention_w48_bias_clobbers.inc" if BIAS else "attention_w48_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_w48.py - do not edit. Registers the assembly body assigns by hand.\n")
        for i in range(0, len(clob), 12):
            f.write(", ".join('"' + c + '"' for c in clob[i:i + 12]) + ("," if i + 12 < len(clob) else "") + "\n")
    n_mfma = sum(1 for ln in lines if "v_mfma" in ln)
    print(f"{len(lines)} lines, {n_mfma} MFMAs -> {os.path.normpath(out)}")


if __name__ == "__main__":
    main()
