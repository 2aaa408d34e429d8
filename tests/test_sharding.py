"""How the units are dealt to the ranks (fetalreconstruction_amd/sharding.py shard_units; csrc/svr_shard.h spatial_order does the same in
C++ for the two command lines): every unit once, ranges in rank order, balanced by work, and -- layout "spatial" -- a rank's units are a
contiguous segment of EVERY stack (neighbours in space)."""
import numpy as np
import pytest

from fetalreconstruction_amd.sharding import shard_units


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("layout", ["spatial", "contiguous"])
def test_every_unit_once_and_balanced(world, layout):
    rng = np.random.default_rng(world)
    counts = [70, 64, 70, 31, 9]
    si = np.repeat(np.arange(len(counts)), counts)
    work = np.concatenate([np.sin(np.linspace(0.05, 3.1, c)) * rng.uniform(0.5, 1.0, c) * (1 + 0.3 * k) for k, c in enumerate(counts)])
    order, ranges = shard_units(work, si, world, layout)
    assert sorted(order.tolist()) == list(range(len(work)))
    assert ranges[0][0] == 0 and ranges[-1][1] == len(work) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    tot = np.array([work[order[lo:hi]].sum() for lo, hi in ranges])
    assert tot.max() <= tot.mean() + 1.01 * work.max()               # within about one unit's work of the mean
    if layout == "contiguous":
        assert np.array_equal(order, np.arange(len(work)))
        return
    for lo, hi in ranges:
        mine = order[lo:hi]
        assert np.all(np.diff(si[mine]) >= 0)                         # inside a rank: stack after stack ...
        for s in np.unique(si[mine]):
            seg = mine[si[mine] == s]
            assert np.array_equal(seg, np.arange(seg[0], seg[0] + len(seg)))      # ... and one contiguous segment of each
    if world > 1:
        # rank r's segment of a stack lies before rank r + 1's
        for s in range(len(counts)):
            firsts = [order[lo:hi][si[order[lo:hi]] == s] for lo, hi in ranges]
            last = -1
            for seg in firsts:
                if len(seg):
                    assert seg[0] > last
                    last = seg[-1]


def test_the_command_lines_deal_the_units_the_same_way(tmp_path):
    """csrc/svr_shard.h spatial_order against sharding.shard_units on the same work vector, through a 20-line C++ program"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "so.cpp"
    src.write_text('''#include <cstdio>
#include <vector>
#include "%s/fetalreconstruction_amd/csrc/svr_shard.h"
int main(int argc, char **argv) {
  int n, world; if (scanf("%%d %%d", &n, &world) != 2) return 1;
  std::vector<double> w(n); std::vector<int> st(n);
  for (int i = 0; i < n; ++i) if (scanf("%%lf %%d", &w[i], &st[i]) != 2) return 1;
  std::vector<int> order, lo, hi;
  svr::spatial_order(w, st, world, order, lo, hi);
  for (int r = 0; r < world; ++r) printf("%%d %%d\\n", lo[r], hi[r]);
  for (int k : order) printf("%%d\\n", k);
  return 0;
}''' % root)
    exe = tmp_path / "so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", str(exe), str(src)])
    rng = np.random.default_rng(5)
    counts = [40, 33, 57]
    si = np.repeat(np.arange(3), counts)
    work = rng.uniform(0.0, 1000.0, si.size).round(3)
    work[rng.integers(0, si.size, 12)] = 0.0                         # slices outside the mask
    for world in (2, 3, 8):
        order, ranges = shard_units(work, si, world, "spatial")
        inp = f"{si.size} {world}\n" + "\n".join(f"{w:.3f} {s}" for w, s in zip(work, si)) + "\n"
        out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.split("\n")
        got_ranges = [tuple(int(v) for v in out[r].split()) for r in range(world)]
        got_order = [int(v) for v in out[world:] if v.strip()]
        assert got_ranges == [tuple(map(int, r)) for r in ranges] and got_order == order.tolist()
