"""The halo-staged conv kernel (csrc/conv_halo.inc) orders its LDS-DMA loads against its fragment reads with counted `s_waitcnt vmcnt(N)`
immediates derived by hand in the file's header. This test replays the kernel's issue order - prologue, first triple, steady triples,
tail - for both instances (2 or 1 weight pieces per wave and K-tile) and several channel counts, and checks on the model what the
assembly generators' checkers prove on their streams: at every mid-tile wait the loads the NEXT K-tile reads (its weight tile; its slab if
it opens a triple) are older than the N youngest loads in flight, no slab / weight slot is overwritten while a K-tile that reads it is
still ahead of the barrier that precedes the overwrite (and its fragment reads have RETIRED before that barrier: an lgkmcnt(0) in the
wait - round-3 advice), and every load issued is waited for before the epilogue. The immediates and ring
sizes are read from the source, so an edit there that breaks the schedule fails here without a GPU.
(VideoConvolution.swift:202-348 is the conv being computed; the schedule itself has no reference counterpart.)"""
import os
import re

import pytest

DX2_WAIT_RETIRES_LDS_READS = None

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-video-swift-mlx_amd", "csrc", "conv_halo.inc")


def _constants():
    text = open(SRC).read()
    m = re.search(r"NS = (\d+), NB = (\d+)", text)
    ns, nb = int(m.group(1)), int(m.group(2))
    m2 = re.search(r"if constexpr \(DX == 2\) (wait_\w+)<(\d+) \* BPW>\(\);", text)
    m = re.search(r"else wait_vmcnt_barrier<FIRST \? (\d+) \* BPW : (\d+) \* BPW \+ (\d+)>", text)
    assert m and m2, "steady wait expressions changed: update this model"
    tight_b, loose_b, loose_c = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert int(m2.group(2)) == tight_b
    global DX2_WAIT_RETIRES_LDS_READS
    DX2_WAIT_RETIRES_LDS_READS = m2.group(1) == "wait_lgkm_vmcnt_barrier"  # lgkmcnt(0) in the wait that precedes the slab re-stage
    assert re.search(r"wait_vmcnt_barrier<BPW \* \(PDB - 1\)>", text), "prologue wait changed: update this model"
    assert "if constexpr (DX == 2) stage_slab(sbuf);" in text and "stage_slab(0);\n    stage_slab(1);" in text
    return ns, nb, tight_b, loose_b, loose_c


@pytest.mark.parametrize("bpw", [2, 1])
@pytest.mark.parametrize("cpt", [1, 2, 4, 8])
def test_counted_waits_cover_every_fragment_read(bpw, cpt):
    _replay(bpw, cpt, None)


@pytest.mark.parametrize("bpw", [2, 1])
@pytest.mark.parametrize("cpt", [1, 2, 4])
def test_persistent_handover_keeps_the_wait_accounting(bpw, cpt):
    """Round 4: a workgroup walks several tiles; the next tile's first operands are requested from inside the epilogue. The K loop's
    counted waits must hold when it is entered through that handover (fewer loads in flight than after the plain prologue, in the same
    order), and the early request must stay out of the epilogue's scratch slots."""
    _replay(bpw, cpt, None, handover=True)


def test_the_model_rejects_a_slab_restage_without_the_lgkm_wait():
    """Round-3 advice: with a vmcnt-only wait before the DX == 2 barrier, another wave's LDS-DMA may overwrite the slab while this
    wave's fa1 reads are outstanding. The replay must see that."""
    with pytest.raises(AssertionError, match="WAR"):
        _replay(2, 2, False)


def _handover_constants():
    """The persistent kernel's tile handover (round 4), read from the source: how many weight tiles of the NEXT tile are requested before the
    epilogue (PRE_B), where the epilogue's scratch lives, and that the K loop restarts from a clean vmcnt behind a barrier."""
    text = open(SRC).read()
    pre_b = int(re.search(r"constexpr int PRE_B = (\d+);", text).group(1))
    assert "char* smem_epi = smem + B_OFF + PRE_B * B_BYTES;" in text, "epilogue scratch moved: update this model"
    i_epi = text.index("gemm_epilogue<BM, BN, WGM, WGN, (BN == 128), true>(acc, g, em0, en0")
    tail = text[i_epi:]
    i_wait = tail.index('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        __syncthreads();')
    i_rest = tail.index("for (int q = PRE_B; q < PDB; ++q) stage_b();")
    assert 0 < i_wait < i_rest, "the remaining weight tiles must be staged behind vmcnt(0) + barrier"
    assert "for (int q = 0; q < PRE_B; ++q) stage_b();" in text
    return pre_b


def _replay(bpw, cpt, dx2_lgkm, handover=False):
    ns, nb, tight_b, loose_b, loose_c = _constants()
    if dx2_lgkm is None:
        dx2_lgkm = DX2_WAIT_RETIRES_LDS_READS
    assert ns == 2
    pdb = nb - 1
    nq, nk = 9 * cpt, 27 * cpt
    issued = []            # loads in issue order: ("slab", triple) x 4 or ("w", tile) x bpw
    slab_of_buf = {}       # buffer -> triple whose slab was staged there last
    tile_of_slot = {}      # weight ring slot -> K-tile staged there last

    def stage_slab(q, buf):
        issued.extend([("slab", q)] * 4)
        slab_of_buf[buf] = q

    def stage_w(t):
        issued.extend([("w", t)] * bpw)
        tile_of_slot[t % nb] = t

    landed_early = set()
    retired = []           # loads a vmcnt(0) has already retired (handover), kept for the exactly-once bookkeeping

    def landed_after_wait(n):
        return (set(issued[:len(issued) - n]) if n else set(issued)) | landed_early

    if not handover:
        # prologue of a workgroup's first tile
        stage_slab(0, 0)
        stage_slab(1, 1)
        for t in range(pdb):
            stage_w(t)
    else:
        # a later tile of a persistent workgroup: slabs 0 / 1 and the weights of K-tiles 0 .. PRE_B - 1 were requested from inside the
        # previous tile's epilogue - into slabs 0 / 1 and ring slots 0 .. PRE_B - 1, while the epilogue's scratch occupies slots PRE_B .. -
        # then vmcnt(0) + barrier, then the remaining weight tiles
        pre_b = _handover_constants()
        assert 1 <= pre_b <= pdb and nb - pre_b >= 2, "scratch needs the slots the early request leaves free"
        stage_slab(0, 0)
        stage_slab(1, 1)
        for t in range(pre_b):
            stage_w(t)
        assert all(slot < pre_b for slot in tile_of_slot), "an early weight request lands in the epilogue's scratch"
        issued_before = list(issued)
        retired.extend(issued_before)
        del issued[:]            # vmcnt(0): nothing of it is in flight any more ...
        landed_early |= set(issued_before)
        for t in range(pre_b, pdb):
            stage_w(t)
    ok = landed_after_wait(bpw * (pdb - 1))
    assert ("slab", 0) in ok and ("w", 0) in ok
    next_slab = 2
    drained = False
    for t in range(nk):
        q, dx = divmod(t, 3)
        steady = t < nk - pdb
        first = q == 0
        # first half reads K-tile t's second fragments: its slab and weights must be the ones in LDS
        assert slab_of_buf[q % ns] == q, (t, slab_of_buf)
        assert tile_of_slot[t % nb] == t, (t, tile_of_slot)
        # the first half has ISSUED the second fragments' reads (fa1 / fb1); the MFMAs that consume them sit behind the barrier, so only an
        # lgkmcnt(0) in the wait retires them before it (the first fragments, read in the previous second half, were consumed by this
        # first half's MFMAs and are retired)
        reads_in_flight = {("slab", q % ns), ("w", t % nb)}
        # mid-tile wait + barrier
        if steady:
            n = tight_b * bpw if (dx == 2 or first) else loose_b * bpw + loose_c
            ok = landed_after_wait(n)
            if dx == 2 and dx2_lgkm:
                reads_in_flight = set()
        elif not drained:
            ok = landed_after_wait(0)
            drained = True
        if t + 1 < nk:
            assert ("w", t + 1) in ok, (t, "weights of the next K-tile may still be in flight")
            if dx == 2:
                assert ("slab", q + 1) in ok, (t, "the next triple's slab may still be in flight")
        # second half: staging behind the barrier, then the first fragments of K-tile t + 1
        if steady:
            if dx == 2:
                # the buffer being overwritten belongs to triple q, whose last fragment read was in this iteration's first half
                assert slab_of_buf[q % ns] == q
                assert ("slab", q % ns) not in reads_in_flight, (t, "slab re-staged while fragment reads of it may be outstanding (WAR)")
                stage_slab(next_slab, q % ns)
                next_slab += 1
            # the slot being overwritten held K-tile t - 1 (read in the previous iteration, a barrier ago)
            slot = (t + pdb) % nb
            assert tile_of_slot.get(slot, -1) in (-1, t - 1) or t == 0, (t, slot, tile_of_slot)
            assert ("w", slot) not in reads_in_flight, (t, "weight slot re-staged while fragment reads of it may be outstanding (WAR)")
            stage_w(t + pdb)
        if t + 1 < nk:
            q1 = (t + 1) // 3
            assert slab_of_buf[q1 % ns] == q1 and tile_of_slot[(t + 1) % nb] == t + 1
    assert next_slab == nq, "every triple's slab is staged exactly once"
    assert sorted(set(x[1] for x in retired + issued if x[0] == "w")) == list(range(nk)), "every K-tile's weights are staged exactly once"
    assert drained, "the tail drains the loads in flight before the epilogue reuses LDS"


def test_the_model_rejects_a_wait_that_is_too_loose():
    """The same replay with the steady immediate raised by one piece must fail - the check is not vacuous."""
    ns, nb, tight_b, loose_b, loose_c = _constants()
    bpw, cpt = 2, 2
    pdb = nb - 1
    nk = 27 * cpt
    issued = []
    for _ in range(2):
        issued.extend([("slab", _)] * 4)
    for t in range(pdb):
        issued.extend([("w", t)] * bpw)
    bad = False
    next_slab = 2
    for t in range(nk - pdb):
        q, dx = divmod(t, 3)
        n = (tight_b * bpw if (dx == 2 or q == 0) else loose_b * bpw + loose_c) + bpw  # one weight tile too many allowed in flight
        ok = set(issued[:len(issued) - n])
        if ("w", t + 1) not in ok:
            bad = True
            break
        if dx == 2:
            issued.extend([("slab", next_slab)] * 4)
            next_slab += 1
        issued.extend([("w", t + pdb)] * bpw)
    assert bad
