"""TEST INFRASTRUCTURE: a plain restatement of the reference driver's junction stitching (src/python/segment.py:157-165,199-252),
used by the test-suite as the checker of the product's native stitcher (csrc/stitch.h) — on border lists produced by any chunk
engine (the oracle, adversarial engines, the HIP path).  Pinned by vectors captured from the reference's own Python
(tests/golden/driver_cases.json: `funcs`, `increase_patch`, and the whole-driver cases).  Not imported by the product.

  repeated(b1, b2)       mask over b1 ++ b2 of the values that occur more than once       (the reference's find_dups: a pandas
                          duplicated(keep=False) on the concatenation)
  shared(b1, b2)         how many entries of b1 ++ b2 are repeated                          (is_2_overlap)
  splice(b1, b2)         b1 up to its first entry that recurs, then what b2 holds beyond it  (merge2)
  next_patch(size, cap)  patch growth rule                                                   (increase_patch)
  join(b1, b2, many)     one junction, patches from `many([(start, end)]) -> [borders]`      (stitch_2_dfs)
  tree(chunks, many)     pairwise reduction (0,1),(2,3),... until one list is left           (merge_df_list)
"""
import numpy as np


class StitchError(ValueError):
    pass


MSG_NOT_ADJACENT = '[wt segment] Patch stitching Failed!              patches are not supposed to be merged'
MSG_GAVE_UP = '[wt segment] Patch stitching Failed!              Try increasing chunk size (--chunk_size flag)'


def repeated(b1, b2):
    both = np.concatenate([np.asarray(b1), np.asarray(b2)])
    order = np.argsort(both, kind='stable')
    srt = both[order]
    same_next = np.zeros(srt.size, dtype=bool)
    same_next[:-1] = srt[1:] == srt[:-1]
    rep_sorted = same_next.copy()
    rep_sorted[1:] |= same_next[:-1]
    out = np.empty(srt.size, dtype=bool)
    out[order] = rep_sorted
    return out


def shared(b1, b2):
    return int(repeated(b1, b2).sum())


def splice(b1, b2):
    b1, b2 = np.asarray(b1), np.asarray(b2)
    first = int(np.flatnonzero(repeated(b1, b2))[0])                # position in b1 ++ b2 of the first repeated entry
    pivot = b1[first]
    return np.concatenate([b1[:first + 1], b2[np.searchsorted(b2, pivot) + 1:]])


def next_patch(size, cap):
    return cap + 1 if size == cap else int(min(2 * size, cap))


def join(b1, b2, many, error=StitchError):
    b1, b2 = np.asarray(b1), np.asarray(b2)
    if b1[-1] != b2[0]:
        raise error(MSG_NOT_ADJACENT)
    cut = int(b1[-1])
    room = (int(b1[-1] - b1[0]), int(b2[-1] - b2[0]))
    reach = [min(50, room[0]), min(50, room[1])]
    while reach[0] <= room[0] and reach[1] <= room[1]:
        patch = np.asarray(many([(cut - reach[0], cut + reach[1])])[0])
        left, right = shared(b1, patch), shared(patch, b2)
        if left and right:
            return splice(splice(b1, patch), b2)
        if not left:
            reach[0] = next_patch(reach[0], room[0])
        if not right:
            reach[1] = next_patch(reach[1], room[1])
    raise error(MSG_GAVE_UP)


def tree(chunks, many, error=StitchError):
    level = [np.asarray(c) for c in chunks]
    while len(level) > 1:
        paired = [join(level[i - 1], level[i], many, error) for i in range(1, len(level), 2)]
        level = paired + ([level[-1]] if len(level) % 2 else [])
    return level[0]
