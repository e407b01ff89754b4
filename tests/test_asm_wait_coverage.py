"""Static wait-coverage proof of the generated gfx950 assembly (tools/gen_attn_w48.py check_wait_coverage): every ds_read of an
LDS ring slot is preceded by a vmcnt wait that retires the slot's LDS-DMA fills AND a barrier; every refill is preceded by an
lgkmcnt wait + barrier behind the slot's last reads. Runs on CPU (no GPU, no assembler): it walks the instruction text.

Background: a prologue that left two staged tiles in flight (`vmcnt(16)`) once passed every parity test and broke run-to-run
repeatability at full size about every second forward. The checker must reject exactly that stream.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "gen_attn_w48.py")


@pytest.mark.parametrize("variant", [[], ["--bias"], ["--prescaled"], ["--bias", "--prescaled"]])
def test_attention_stream_is_proven_and_matches_the_committed_file(variant):
    r = subprocess.run([sys.executable, GEN, "--check"] + variant, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wait coverage ok" in r.stdout and "matches" in r.stdout, r.stdout


@pytest.mark.parametrize("variant", [[], ["--bias"], ["--prescaled"], ["--bias", "--prescaled"]])
def test_checker_rejects_the_prologue_race_that_once_shipped(variant):
    r = subprocess.run([sys.executable, GEN, "--check", "--inject-prologue-race"] + variant, capture_output=True, text=True)
    assert r.returncode != 0
    assert "WaitCoverageError" in r.stderr and "RAW: ds_read of (1, 'K')" in r.stderr, r.stderr[-600:]


def test_checker_rejects_a_refill_without_a_barrier_behind_the_reads():
    """WAR direction, on a hand-made stream: slot 0 is read, then refilled with no barrier in between."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    argv = sys.argv
    sys.argv = ["x"]
    try:
        import gen_attn_w48 as g
    finally:
        sys.argv = argv
    fill = ["s_add_u32 m0, %[wlds], 0", "s_nop 0", "buffer_load_dwordx4 %[ko0], s[36:39], s44 offen lds"]
    read = ["ds_read_b128 v[168:171], %[ka0] offset:0"]
    ok = fill + ["s_waitcnt vmcnt(0)", "s_barrier", "10:"] + read + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + fill + \
        ["s_waitcnt vmcnt(0)", "s_barrier", "s_branch 10b"]
    assert g.check_wait_coverage(ok)["ring_reads"] == 3
    bad = fill + ["s_waitcnt vmcnt(0)", "s_barrier", "10:"] + read + ["s_waitcnt lgkmcnt(0)"] + fill + \
        ["s_waitcnt vmcnt(0)", "s_barrier", "s_branch 10b"]
    with pytest.raises(g.WaitCoverageError, match="WAR"):
        g.check_wait_coverage(bad)
    early = fill + ["s_barrier", "s_waitcnt vmcnt(0)", "10:"] + read + ["s_branch 10b"]   # wait AFTER the barrier: not visible
    with pytest.raises(g.WaitCoverageError, match="RAW"):
        g.check_wait_coverage(early)


DTL = os.path.join(ROOT, "tools", "gen_gemm_asm_dtl.py")


def test_gemm_192x256_stream_is_proven_and_matches_the_committed_file():
    r = subprocess.run([sys.executable, DTL, "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "matches" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("flag,kind", [("--inject-raw-race", "RAW"), ("--inject-war-race", "WAR")])
def test_gemm_192x256_checker_rejects_injected_races(flag, kind):
    r = subprocess.run([sys.executable, DTL, "--check", flag], capture_output=True, text=True)
    assert r.returncode != 0 and "WaitCoverageError: " + kind in r.stderr, r.stderr[-400:]


X32 = os.path.join(ROOT, "tools", "gen_attn_x32.py")


def test_attention_x32_stream_is_proven_and_matches_the_committed_file():
    """The 32x32x16 stream keeps fragment reads in flight ACROSS the step barrier and ends every step with vmcnt(0): the proof is
    what says that the reads issued ahead only ever touch slots that were visible one barrier earlier."""
    r = subprocess.run([sys.executable, X32, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wait coverage ok" in r.stdout and "matches" in r.stdout, r.stdout


@pytest.mark.parametrize("flag,kind", [("--inject-raw-race", "RAW"), ("--inject-war-race", "WAR")])
def test_attention_x32_checker_rejects_injected_races(flag, kind):
    r = subprocess.run([sys.executable, X32, "--check", flag], capture_output=True, text=True)
    assert r.returncode != 0 and "WaitCoverageError: " + kind in r.stderr, r.stderr[-400:]
