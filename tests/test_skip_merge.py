"""The arithmetic claim behind skip_to_box / skip_empty_block (zig_vulkan_amd/csrc/vrt_trace.hip), checked on the CPU in binary32:

the shader's brick-level walk (comp:345-372) is a three-way merge of the side-distance sequences of x, y and z — each built by
repeated addition of |1/dir| — with ties going to z, then y, then x.  Therefore the walk's state right after the need-th crossing
of an axis A is: T = A's side distance after need-1 additions; every other axis has consumed exactly its elements that precede T
in merge order (c < T, or c == T if the axis wins the tie against A); A's side distance is T + |1/dir_A|.

This file walks step by step (the shader's branchy selection, as the oracle does) and compares with the closed form, including
rays built to produce ties (equal direction components, lattice start positions)."""
import numpy as np
import pytest

F = np.float32
PRIORITY = {0: 0, 1: 1, 2: 2}  # x < y < z: the higher one is crossed first among equal side distances


def _walk_until(sd, d, axis, need, limit=100000):
    """comp:345-372: x<y ? (x<z ? X : Z) : (y<z ? Y : Z), side_dist[a] += delta[a]; until `axis` has been crossed `need` times.
    Returns side distances, crossings per axis, and the crossed distance of the last step."""
    sd = [F(v) for v in sd]
    n = [0, 0, 0]
    t = F(0)
    for _ in range(limit):
        if sd[0] < sd[1]:
            a = 0 if sd[0] < sd[2] else 2
        else:
            a = 1 if sd[1] < sd[2] else 2
        t = sd[a]
        sd[a] = F(sd[a] + d[a])
        n[a] += 1
        if a == axis and n[a] == need:
            return sd, n, t
    raise AssertionError("axis never crossed")


def _closed_form(sd, d, axis, need):
    t = F(sd[axis])
    for _ in range(need - 1):
        t = F(t + d[axis])
    out, n = [None] * 3, [0, 0, 0]
    for b in range(3):
        if b == axis:
            out[b], n[b] = F(t + d[b]), need
            continue
        c, k = F(sd[b]), 0
        wins_tie = PRIORITY[b] > PRIORITY[axis]
        while (c <= t) if wins_tie else (c < t):
            c = F(c + d[b])
            k += 1
        out[b], n[b] = c, k
    return out, n, t


def _ray(rng, ties):
    if ties:
        # direction components from a tiny set, start on a coarse lattice: many equal side distances
        dirc = rng.choice([0.25, 0.5, 0.5, 1.0, 1.0, 0.75], 3) * rng.choice([-1.0, 1.0], 3)
        frac = rng.integers(0, 5, 3) / 4.0
    else:
        v = rng.normal(size=3)
        dirc = v / np.linalg.norm(v)
        frac = rng.random(3)
    d = [F(abs(F(1.0) / F(c))) for c in dirc]
    sd = [F(F(f) * d[i]) for i, f in enumerate(frac)]
    return sd, d


@pytest.mark.parametrize("ties", [False, True])
def test_state_after_the_nth_crossing_is_the_merge_of_the_three_sequences(ties):
    rng = np.random.default_rng(11 if ties else 7)
    for _ in range(1500):
        sd, d = _ray(rng, ties)
        axis = int(rng.integers(0, 3))
        need = int(rng.integers(1, 40))
        want_sd, want_n, want_t = _walk_until(sd, d, axis, need)
        got_sd, got_n, got_t = _closed_form(sd, d, axis, need)
        assert got_n == want_n, (sd, d, axis, need)
        assert [np.float32(v).view(np.uint32) for v in got_sd] == [np.float32(v).view(np.uint32) for v in want_sd]
        assert F(got_t).view(np.uint32) == F(want_t).view(np.uint32)


def test_exit_of_a_4x4x4_block_is_the_first_exit_crossing_in_merge_order():
    """skip_empty_block: each axis has its exit crossing (the e-th, e = cells to the block's face + 1); the walk leaves the block
    with the one that comes first in merge order — smallest distance, z before y before x among equals."""
    rng = np.random.default_rng(3)
    for it in range(1500):
        sd, d = _ray(rng, it % 2 == 0)
        e = [int(v) for v in rng.integers(1, 5, 3)]           # crossings until the block's face, per axis
        # step by step until some axis reaches its count
        s, n = [F(v) for v in sd], [0, 0, 0]
        while True:
            if s[0] < s[1]:
                a = 0 if s[0] < s[2] else 2
            else:
                a = 1 if s[1] < s[2] else 2
            t = s[a]
            s[a] = F(s[a] + d[a])
            n[a] += 1
            if n[a] == e[a]:
                break
        # closed form: exit distances, first in merge order, then the merge
        tx = []
        for b in range(3):
            v = F(sd[b])
            for _ in range(e[b] - 1):
                v = F(v + d[b])
            tx.append(v)
        if tx[2] <= tx[0] and tx[2] <= tx[1]:
            ax = 2
        elif tx[1] <= tx[0]:
            ax = 1
        else:
            ax = 0
        assert ax == a, (sd, d, e)
        got_sd, got_n, got_t = _closed_form(sd, d, ax, e[ax])
        assert got_n == n and F(got_t).view(np.uint32) == F(t).view(np.uint32)
        assert [np.float32(v).view(np.uint32) for v in got_sd] == [np.float32(v).view(np.uint32) for v in s]
        assert all(got_n[b] < e[b] for b in range(3) if b != ax)   # the others are still inside: at most three elements each


def test_strict_comparison_as_a_non_strict_one_against_the_next_float_below():
    """The merge loop tests `c <= lim` only; an axis that loses ties gets lim = next_below(T) (bit pattern - 1 for T > 0,
    + 1 for T < 0, -denorm_min for T == 0): c < T  <=>  c <= next_below(T) for every finite c."""
    def next_below(t):
        b = np.float32(t).view(np.uint32)
        if t > 0:
            return np.uint32(b - 1).view(np.float32)
        if t < 0:
            return np.uint32(b + 1).view(np.float32)
        return np.uint32(0x80000001).view(np.float32)
    rng = np.random.default_rng(1)
    ts = np.concatenate([rng.normal(size=200).astype(np.float32) * 50, np.float32([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, 3.4e38])])
    for t in ts:
        nb = next_below(t)
        assert nb == np.nextafter(np.float32(t), np.float32(-np.inf), dtype=np.float32) or (t == 0 and nb < 0)
        for c in [np.nextafter(t, np.float32(-np.inf), dtype=np.float32), t, np.nextafter(t, np.float32(np.inf), dtype=np.float32), np.float32(0), np.float32(-1e30), np.float32(1e30)]:
            assert (c < t) == (c <= nb), (t, c)
