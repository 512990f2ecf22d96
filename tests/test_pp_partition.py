"""The balanced split's work list (csrc/conv_pp.hip, PpWork / segment()), restated in Python and checked for what the
kernel relies on, over many launch sizes: every k-step pair of every tile is owned exactly once; a cut tile's pieces
are owned by consecutive workgroups of one XCD group in k order; a workgroup produces at most one slab and consumes at
most one (its predecessor's); pieces are at least four k-steps long; pieces from the middle of a tile only occur with a
whole round before them.  CPU only: the kernel itself is compared bit for bit with the unsplit one in test_gpu_conv.py."""
import pytest

FULL, HEAD, TAIL, MID = 0, 1, 2, 3


def lock_rounds(nblk, grid):
    q, G = nblk >> 3, grid >> 3
    return (2 * q - G) // (2 * G)


def work_list(block, grid, total, nk, rl):
    """[(tile j within the XCD group's range, kt0, kt1, kind)] in execution order, as segment() enumerates it."""
    x, i, G = block & 7, block >> 3, grid >> 3
    q, r = total >> 3, total & 7
    ln = q + (1 if x < r else 0)
    npt = nk >> 1
    U = (ln - rl * G) * npt

    def cut(k):
        u = (k * U) // G
        o = u % npt
        if o < 2:
            u -= o
        elif o > npt - 2:
            u += npt - o
        return u
    u0, u1 = cut(i), cut(i + 1)
    any_ = u1 > u0
    ja = u0 // npt
    oa2 = 2 * (u0 - ja * npt) if any_ else 0
    jb = (u1 - 1) // npt if any_ else ja
    ob2 = 2 * (u1 - jb * npt) if any_ else nk
    has_mid = any_ and ja == jb and oa2 > 0 and ob2 < nk
    has_tail = oa2 > 0 and not has_mid
    has_head = any_ and ob2 < nk and not has_mid
    mid_at = 1 if rl > 0 else 0
    first_full = ja + has_tail
    last_full = jb - has_head
    n_full = last_full - first_full + 1 if (any_ and not has_mid and last_full >= first_full) else 0
    nseg = rl + has_head + has_mid + n_full + has_tail
    out = []
    for sidx in range(nseg):
        s = sidx
        if has_head:
            if s == 0:
                out.append((rl * G + jb, 0, ob2, HEAD))
                continue
            s -= 1
        if has_mid:
            if s == mid_at:
                out.append((rl * G + ja, oa2, ob2, MID))
                continue
            if s > mid_at:
                s -= 1
        if s < rl:
            out.append((i + s * G, 0, nk, FULL))
        elif s - rl < n_full:
            out.append((rl * G + first_full + s - rl, 0, nk, FULL))
        else:
            out.append((rl * G + ja, oa2, nk, TAIL))
    return ln, out


@pytest.mark.parametrize("nk", [4, 6, 8, 16, 18, 36, 72])
def test_balanced_split_partition_covers_every_unit_once(nk):
    grid = 256
    for total in list(range(257, 700)) + [722, 903, 1444, 1445, 2049, 5776, 23105]:
        if total % grid == 0:
            continue
        rl = lock_rounds(total, grid)
        assert rl >= 0
        for x in range(8):
            owned, lists = {}, []
            for i in range(grid >> 3):
                ln, wl = work_list(i * 8 + x, grid, total, nk, rl)
                lists.append(wl)
                cons = [s for s in wl if s[3] in (TAIL, MID)]
                prod = [s for s in wl if s[3] in (HEAD, MID)]
                assert len(cons) <= 1 and len(prod) <= 1
                for k, (j, kt0, kt1, kind) in enumerate(wl):
                    assert 0 <= j < ln and kt1 - kt0 >= 4 and kt0 % 2 == 0 and kt1 % 2 == 0
                    if kind == MID:
                        assert rl >= 1 and k == 1 and wl[0][3] == FULL
                    if kind == HEAD:
                        assert k == 0
                    if kind == TAIL:
                        assert k == len(wl) - 1
                    for kt in range(kt0, kt1, 2):
                        assert (j, kt) not in owned
                        owned[(j, kt)] = i
                if cons:            # the k-steps before this piece: the predecessor's one producing piece, same tile, meeting at kt0
                    assert i > 0
                    pp = [s for s in lists[i - 1] if s[3] in (HEAD, MID)]
                    assert len(pp) == 1 and pp[0][0] == cons[0][0] and pp[0][2] == cons[0][1]
                if prod:            # ... and somebody continues what this one leaves
                    assert i + 1 < (grid >> 3)
            assert len(owned) == ln * (nk >> 1)
