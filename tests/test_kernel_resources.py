"""No kernel of the product may spill a vector register or touch scratch memory: a spilled
loop-carried value costs HBM traffic in the hot loop (round 1 found +24 GB per launch that way).  hipcc cross-compiles
gfx950 without a GPU; the table comes from -Rpass-analysis=kernel-resource-usage with the build's own flags
(tools/kernel_resources.py)."""
import concurrent.futures
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SOURCES = ["gt4hip_kernels.hip", "gt4hip_nway.hip", "gt4hip_sort.hip", "gt4hip_api.hip"]  # every file of csrc that holds a __global__ function


@pytest.fixture(scope="module")
def tables():
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not on PATH")
    import kernel_resources as K
    with concurrent.futures.ThreadPoolExecutor(len(SOURCES)) as ex:
        return dict(zip(SOURCES, ex.map(K.table, SOURCES)))


def test_no_kernel_spills_vector_registers_or_uses_scratch(tables):
    bad = [(src, r["name"], r["vspill"], r["scratch"]) for src, rows in tables.items() for r in rows if r["vspill"] != 0 or r["scratch"] != 0]
    assert not bad, "kernels with VGPR spills / scratch: %s" % bad


def test_every_instantiation_of_the_merge_kernels_is_seen(tables):
    """the table is not empty by accident: the pair kernel's three geometries and the N-way modes are all there"""
    names = [r["name"] for r in tables["gt4hip_kernels.hip"]]
    for inst in ("k_pair_merge<1024, 6, 1, 2, 1, 0>", "k_pair_merge<1024, 4, 1, 0, 1, 5>", "k_pair_merge<1024, 4, 1, 0, 0, 0>", "k_pair_merge<512, 4, 0, 0, 0, 0>"):
        assert inst in names, inst
    nway = [r["name"] for r in tables["gt4hip_nway.hip"]]
    # the N-way sources are compiled twice (round 5): eight lists per launch, thirty-two -- five modes each (union, count,
    # merged samples, the two count tables)
    assert sum("km8::k_nway_merge<" in n for n in nway) == 5, nway
    assert sum("km32::k_nway_merge<" in n for n in nway) == 5, nway


def test_launch_bounds_hold(tables):
    """register counts stay inside what the occupancy the host code sizes its grids for needs: 128 VGPRs at sixteen
    wavefronts per CU (1024 threads), 85 for the count-only geometry's three workgroups of 512"""
    for r in tables["gt4hip_kernels.hip"] + tables["gt4hip_nway.hip"]:
        if r["name"].startswith("k_pair_merge<1024") or "k_nway_merge<1024" in r["name"]:
            assert r["vgpr"] <= 128, r
        if r["name"].startswith("k_pair_merge<512, 4, 0,"):
            assert r["vgpr"] <= 85, r
