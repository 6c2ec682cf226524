"""The rounding bounds of the fp32 screen (score_screen_k) and of the fp32 box tests (cull_tiles32_k), checked on the HOST:
tests/cpp/test_screen_bounds.cpp restates the device arithmetic with fmaf / float operations and runs millions of points --
random scenes over eight orders of magnitude, offsets from the origin up to 1e5 scene sizes, points placed 1e-16 ... 1e-3
thresholds from the cut-off on both sides -- through the screen and through the exact fp64 test.  No GPU needed."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp32_screen_and_box_test_bounds_hold_on_the_host():
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "_build/test_screen_bounds"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_screen_bounds"), "12000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "wrong 0;" in r.stdout


def test_screen_bounds_are_not_slack_by_16():
    """The mutation check of the test above: the same program built with every screen bound (plane, sphere in expanded
    form, cylinder) divided by 16 (M3D_SCREEN_BOUND_DIVISOR) must report wrong verdicts for ALL THREE kinds -- i.e. the
    points placed around the cut-offs really do probe the bounds, and the bounds are within 16x of what is needed."""
    import re
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "_build/test_screen_bounds_div16"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_screen_bounds_div16"), "12000"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "all checks passed" not in r.stdout
    wrong = {int(k): int(w) for k, w in re.findall(r"kind (\d): decided [0-9.]+ % of \d+ points, wrong (\d+)", r.stdout)}
    assert set(wrong) == {0, 1, 2} and all(w > 0 for w in wrong.values()), r.stdout


def test_nn_screen_walk_is_exact_on_the_host():
    """tests/cpp/test_nn_screen.cpp replays sorted_walk32 (the screen of the registration validation's neighbour search: fp32
    arithmetic on 8-byte list entries with 16-bit fixed-point coordinates) with float operations on the host: random grids over nine orders of magnitude, origins at the edge of what the library
    admits, lists with duplicates, reflections (equal distances) and distances 1e-16 apart.  A query the walk calls decided
    must have found the exact fp64 minimum; the rounding bound must hold for every entry; ordinary lists must be decided."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "_build/test_nn_screen"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_nn_screen"), "60000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "wrong 0;" in r.stdout and "bound violations 0," in r.stdout


def test_guided_cut_off_searches_equal_the_unguided_ones():
    """tests/cpp/test_cutoffs.cpp: sphere_cutoffs / cylinder_cutoffs narrow their four bisections around computed guesses
    (r -+ thr and their squares); the results must be the unguided searches' bit for bit on random models over twelve
    orders of magnitude (zero / NaN radii, empty and degenerate intervals included), and the defining property
    `dist(q) < thr <=> lo <= s(q) <= hi` must hold for the values next to both ends."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "_build/test_cutoffs"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_cutoffs"), "30000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "guided != unguided 0;" in r.stdout


def test_plane_histogram_bound_holds_on_the_host():
    """tests/cpp/test_plane_bound.cpp runs plane_pair_ub (misc3d_amd/csrc/m3d_bound_fp.hpp: the code plane_bound_k runs on the
    device, compiled by g++) on random tiles -- plane patches with clutter, exact planes, two planes, clutter only; scenes scaled
    by 1e-2 .. 1e2 and moved up to 1e4 scene sizes from the origin; good frames and arbitrary ones -- against the exact count of
    the reference's fp64 test: the bound must never be below it.  The second build drops the slack and the outward bins
    (M3D_BOUND_NO_SLACK): it must report violations, i.e. the pairs generated really do probe the margins."""
    import re
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "_build/test_plane_bound", "_build/test_plane_bound_no_slack"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_plane_bound"), "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all checks passed" in r.stdout and "violations 0" in r.stdout, r.stdout + r.stderr
    m = subprocess.run([os.path.join(cpp, "_build", "test_plane_bound_no_slack"), "20000"], capture_output=True, text=True, timeout=300)
    assert m.returncode != 0 and "all checks passed" not in m.stdout
    assert int(re.search(r"violations (\d+)", m.stdout).group(1)) > 1000, m.stdout


def test_candidate_cache_certificates_hold_on_the_host():
    """tests/cpp/test_reg_cache.cpp runs the candidate cache's arithmetic (misc3d_amd/csrc/m3d_reg_cache_fp.hpp: the code
    reg_validate_cached_k runs, compiled by g++) against an exact nearest-neighbour search over ALL target points in long double --
    noisy patches, lattices with exact ties, duplicates, clutter; coordinates at 0, 1e3 and 3e5 scene units; scales 1e-3 .. 1e3; poses
    from a tenth of the list's radius to beyond it.  A certified winner must be THE nearest point (an exact tie must not be
    certified), "nothing" must mean nothing within the radius, a bound must stay below the truth.  Two mutations -- the radius
    taken 10 % too large, the rounding bound set to zero -- must be caught."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    for target in ("test_reg_cache", "test_reg_cache_inflate_r", "test_reg_cache_no_slack"):
        subprocess.run(["make", "-C", cpp, "_build/" + target], check=True, capture_output=True)
    r = subprocess.run([os.path.join(cpp, "_build", "test_reg_cache"), "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "violations 0" in r.stdout, r.stdout + r.stderr
    m = re.search(r"winner certified (\d+), nothing (\d+), bounds (\d+)", r.stdout)
    assert m and int(m.group(1)) > 100_000 and int(m.group(2)) > 100 and int(m.group(3)) > 10_000, r.stdout
    for target in ("test_reg_cache_inflate_r", "test_reg_cache_no_slack"):
        r = subprocess.run([os.path.join(cpp, "_build", target), "300"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "violations 0" not in r.stdout, (target, r.stdout)
