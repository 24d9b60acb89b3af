"""Work partition of the persistent LDS-resident mixer (csrc/pw_mlp_lds_kernels.hip), restated on the CPU: workgroup b owns the contiguous
share [G*b/B, G*(b+1)/B) of the (sample, row tile) sequence, walks it sample by sample (re-staging the per-sample weight image where the
sample changes -- a workgroup-uniform decision, so the barriers inside match) and hands the tiles of a segment round-robin to its waves.
Every (sample, tile) must be taken exactly once, by exactly one wave, for any batch, tile count, grid size and wave count."""
import itertools


def partition(N, rps, blocks, waves, rows_per_tile=32):
    Ts = (rps + rows_per_tile - 1) // rows_per_tile
    G = Ts * N
    taken, stagings = {}, 0
    for b in range(blocks):
        g, g_end = G * b // blocks, G * (b + 1) // blocks
        staged = -1
        while g < g_end:
            n = g // Ts
            seg_end = min(g_end, (n + 1) * Ts)
            if staged != n:
                stagings += 1
                staged = n
            for wave in range(waves):
                for gt in range(g + wave, seg_end, waves):
                    key = (n, gt - n * Ts)
                    assert key not in taken, key
                    taken[key] = (b, wave)
            g = seg_end
    return Ts, taken, stagings


def test_every_tile_is_taken_exactly_once():
    for N, rps, blocks, waves in itertools.product((1, 3, 8), (1, 31, 32, 6859, 175616), (1, 7, 256, 512), (8, 12, 16)):
        Ts = (rps + 31) // 32
        blocks = min(blocks, max(1, (Ts * N + waves - 1) // waves))       # the launcher never starts more workgroups than it needs
        Ts, taken, stagings = partition(N, rps, blocks, waves)
        assert len(taken) == Ts * N and set(taken) == {(n, t) for n in range(N) for t in range(Ts)}
        assert stagings <= blocks + N - 1                                  # a share crosses at most the sample boundaries inside it


def test_level1_launch_of_the_headline_batch_is_balanced():
    """8 windows x 56^3 rows on 256 workgroups of 12 waves (64->128->64): 43 904 tiles, 171 or 172 per workgroup, 14 or 15 per wave."""
    Ts, taken, _ = partition(8, 56 ** 3, 256, 12)
    assert Ts == 5488
    per_wave = {}
    for owner in taken.values():
        per_wave[owner] = per_wave.get(owner, 0) + 1
    assert len(per_wave) == 256 * 12 and 13 <= min(per_wave.values()) and max(per_wave.values()) <= 16
