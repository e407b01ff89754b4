"""The tall halo-staged conv kernel (csrc/conv_halo2.inc, round 5) keeps NRT + 2 image rows per (frame tap, channel half) group in a ring of
row slots that rotates from group to group, stages rows and weight tiles behind the mid-tile barrier of fixed positions of a nine-K-tile
group, and orders all of it with counted `s_waitcnt vmcnt(N)` immediates (2 weight pieces + the row pieces of the previous iteration
stay in flight). This test replays the schedule on a model, for all three instances (one image row of 384 voxels / two of 192 / four of 96), several
channel counts and a walk over three tiles (the K loop runs on across tiles: the next tile's first rows and weights are staged by the
current tile's last group), and checks what the assembly generators' checkers prove on their streams:

  * every fragment read finds the RIGHT content in its slot (the row of that group / tile, the weight tile of that K-tile), and the load
    that brought it is older than the loads the preceding mid-tile wait leaves in flight;
  * no slot is overwritten while a K-tile that reads its old content is still ahead, and the old content's last reads were issued before
    a wait that carries lgkmcnt(0) and the barrier behind it;
  * the epilogue's scratch slots are not the target of any load in flight while it runs, and hold none of the next tile's first rows.

The staging table, the slot counts and the wait immediates are read from the source, so an edit there that breaks the schedule fails
here without a GPU. (VideoConvolution.swift:202-348 is the conv being computed; the schedule itself has no reference counterpart.)"""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-video-swift-mlx_amd", "csrc", "conv_halo2.inc")


def _constants(nrt):
    text = open(SRC).read()
    m = re.search(r"return NRT == 1 \? \(([^)]*)\)\s*:\s*NRT == 2 \? \(([^)]*)\)\s*:\s*\(([^)]*)\);", text)
    assert m, "halo2_row_id changed: update this model"
    table = {}
    for q, rid in re.findall(r"q == (\d+) \? (\d+)", m.group({1: 1, 2: 2, 4: 3}[nrt])):
        table[int(q)] = int(rid)
    assert re.search(r"NS = NRT == 1 \? 2 : \(NRT == 2 \? 4 : 8\), RPG = NRT \+ 2;", text) and re.search(r"B_BYTES = BN \* ROW_BYTES, NB = 3;", text)
    ns, rpg, nb = {1: 2, 2: 4, 4: 8}[nrt], nrt + 2, 3
    rsp = (384 // nrt + 2 + 7) // 8
    ppw = (rsp + 7) // 8
    assert '"n"(PREV_ROW ? BPW + PPW : BPW)' in text and "lgkmcnt(0)\\n\\ts_barrier" in text, "mid-tile wait changed: update this model"
    assert "constexpr bool PREV_ROW = halo2_row_id<NRT>((Q + 8) % 9) >= 0;" in text
    # order inside the second half: row, weights, (group hand-over,) next K-tile's fragment reads
    i_row = text.index("if constexpr (ID < RPG) stage_row(slot_of(ID), cur + yc[ID]);")
    i_b = text.index("stage_b();  // weights of K-tile t + 3")
    i_rot = text.index("if constexpr (NQ == 0) gslot0 = (gslot0 + RPG) & (NS - 1);")
    i_rd = text.index("read_frags0(NDY, NDX, nslot);")
    assert i_row < i_b < i_rot < i_rd
    assert "char* smem_epi = smem + ((gslot0 + NS / 2) & (NS - 1)) * RS;" in text
    assert re.search(r"for \(int j = 0; j < NRT; \+\+j\) stage_row\(j, tb\.fb\[0\] \+ tb\.yb\[j\]\);\s*#pragma unroll\s*for \(int t = 0; t < NB; \+\+t\) stage_b\(\);", text)
    return table, ns, rpg, nb, ppw


def _replay(nrt, cpt, ntiles=3, table_override=None, no_lgkm=False):
    table, ns, rpg, nb, ppw = _constants(nrt)
    if table_override is not None:
        table = table_override
    bpw = 2
    ngroups, nk = 3 * cpt, 27 * cpt
    issued = []              # (kind, content, slot) in issue order, one entry per PIECE; content = (tile, group, row j) or (tile, ktile)
    row_slot = [None] * ns   # content of each row slot: (index of its last piece in `issued`, content)
    w_slot = [None] * nb
    done = 0                 # loads with index < done are complete (the last wait's guarantee)
    last_read = {}           # ("r", slot) / ("w", slot) -> (iteration counter of the last fragment read, retired?)
    it = 0                   # global iteration (K-tile) counter across tiles
    gslot0 = 0

    def slot_of(i):
        return i % ns if rpg == ns else (gslot0 + i) % ns

    def stage_row(slot, content):
        key = ("r", slot)
        if key in last_read:
            rd_it, retired = last_read[key]
            assert rd_it < it or (rd_it == it and retired), f"WAR: row slot {slot} restaged in iteration {it}, last read in {rd_it} (retired {retired})"
        for _ in range(ppw):
            issued.append(("row", content, slot))
        row_slot[slot] = (len(issued) - 1, content)

    ws = {"tile": 0, "kt": 0, "slot": 0}

    def stage_b():
        key = ("w", ws["slot"])
        if key in last_read:
            rd_it, retired = last_read[key]
            assert rd_it < it or (rd_it == it and retired), f"WAR: weight slot {ws['slot']} restaged in iteration {it}, last read in {rd_it}"
        for _ in range(bpw):
            issued.append(("w", (ws["tile"], ws["kt"]), ws["slot"]))
        w_slot[ws["slot"]] = (len(issued) - 1, (ws["tile"], ws["kt"]))
        ws["slot"] = (ws["slot"] + 1) % nb
        ws["kt"] += 1
        if ws["kt"] == nk:
            ws["kt"] = 0
            ws["tile"] += 1

    def read(kind, slot, want, retired):
        held = (row_slot if kind == "r" else w_slot)[slot]
        assert held is not None and held[1] == want, f"iteration {it}: {kind} slot {slot} holds {held and held[1]}, wanted {want}"
        assert held[0] < done, f"iteration {it}: {kind} slot {slot} ({want}) read before its load is known complete (piece {held[0]}, done {done})"
        last_read[(kind, slot)] = (it, retired)

    def subs():
        return range(nrt)  # image rows of the tile; every wave reads slot_of(sub + dy)

    # prologue of the first tile
    for j in range(nrt):
        stage_row(j, (0, 0, j))
    for _ in range(nb):
        stage_b()
    bslot = 0
    for tile in range(ntiles):
        more = tile + 1 < ntiles
        done = len(issued)  # vmcnt(0) + barrier at the top of the tile loop
        # k-step-0 fragments of K-tile 0
        for s in subs():
            read("r", slot_of(s + 0), (tile, 0, s), retired=False)
        read("w", bslot, (tile, 0), retired=False)
        for gi in range(ngroups):
            lastg = gi + 1 == ngroups
            for q in range(9):
                dy = q // 3
                t = gi * 9 + q
                # ---- first half: k-step-1 fragments of K-tile t
                for s in subs():
                    read("r", slot_of(s + dy), (tile, gi, s + dy), retired=False)
                read("w", bslot, (tile, t), retired=False)
                # ---- mid-tile wait: leaves the previous iteration's loads in flight; lgkmcnt(0): this wave's reads so far have retired
                prev_row = ((q + 8) % 9) in table
                allowed = bpw + (ppw if prev_row else 0)
                done = max(done, len(issued) - allowed)
                if not no_lgkm:
                    for k in list(last_read):
                        last_read[k] = (last_read[k][0], True)
                # ---- second half: staging, then the k-step-0 fragments of K-tile t + 1
                if q in table:
                    rid = table[q]
                    if rid < rpg:
                        stage_row(slot_of(rid), (tile, gi, rid))
                    else:
                        j = rid - rpg
                        if not lastg:
                            stage_row(slot_of(rid), (tile, gi + 1, j))
                        elif more:
                            stage_row(slot_of(rid), (tile + 1, 0, j))
                        else:
                            stage_row(slot_of(rid), ("dummy", gi, j))
                stage_b()
                nslot = (bslot + 1) % nb
                if q == 8:
                    gslot0 = (gslot0 + rpg) % ns
                if not (lastg and q == 8):
                    ngi, nq = (gi, q + 1) if q < 8 else (gi + 1, 0)
                    for s in subs():
                        read("r", slot_of(s + nq // 3), (tile, ngi, s + nq // 3), retired=False)
                    read("w", nslot, (tile, ngi * 9 + nq), retired=False)
                elif more:
                    # the prefetch of the next tile's first K-tile (its values are read again at the top of the tile loop): must be valid LDS content
                    for s in subs():
                        read("r", slot_of(s), (tile + 1, 0, s), retired=False)
                    read("w", nslot, (tile + 1, 0), retired=False)
                bslot = nslot
                it += 1
        # ---- epilogue: scratch = NS / 2 slots from (gslot0 + NS / 2) % NS, contiguous
        first = (gslot0 + ns // 2) % ns
        assert first + ns // 2 <= ns, f"epilogue scratch wraps around the slot ring (gslot0 {gslot0})"
        scratch = set(range(first, first + ns // 2))
        if more:
            nxt = {slot_of(j) for j in range(nrt)}
            assert not (scratch & nxt), "epilogue scratch overlaps the next tile's first rows"
        for i in range(done, len(issued)):  # loads still in flight when the K loop ends
            kind, _, slot = issued[i]
            assert kind == "w" or slot not in scratch, f"a row load into scratch slot {slot} is in flight during the epilogue"
        # every wave has passed the barrier that ends the K loop with its reads retired (the epilogue's own LDS traffic follows)
        for k in list(last_read):
            last_read[k] = (last_read[k][0], True)
    return len(issued)


@pytest.mark.parametrize("nrt,cpt", [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (2, 4), (4, 2), (4, 4), (4, 8)])
def test_tall_kernel_schedule_is_covered_by_its_counted_waits(nrt, cpt):
    _replay(nrt, cpt)


def test_an_odd_number_of_channel_halves_is_refused_for_four_rows():
    """NRT = 4 rotates six rows through eight slots: with an odd number of groups per tile the epilogue's four scratch slots would wrap
    around the ring at every other tile - conv_halo2_takes requires C % 128 == 0 there, and the model shows why."""
    with pytest.raises(AssertionError, match="wraps"):
        _replay(4, 1)
    takes = open(os.path.join(os.path.dirname(SRC), "gemm.hip")).read()
    assert "if (nrt == 4 && q.C % 128 != 0) return 0;" in takes


def test_the_model_rejects_broken_schedules():
    """The replay must see (a) a row staged too late for its first reader, (b) a row staged into a slot whose readers are still ahead,
    (c) a restage whose slot was read in the same iteration without the lgkmcnt(0) in the wait."""
    with pytest.raises(AssertionError, match="read before its load"):
        _replay(2, 2, table_override={1: 2, 2: 4, 3: 3, 5: 5})     # row 2 one K-tile later: its first reader is the prefetch in q = 2
    with pytest.raises(AssertionError, match="holds"):
        _replay(2, 2, table_override={0: 2, 1: 4, 3: 3, 5: 5})     # the next group's row 0 while K-tile 2 still has to read slot 0: wrong content
    with pytest.raises(AssertionError, match="WAR"):
        _replay(2, 2, no_lgkm=True)
