"""Host-side mirror of the library's counter-based generator (Threefry-2x32-13) for the few
per-image SCALAR draws the host makes (motion-blur angle, frost crop origin, ...), plus the
process-wide seed / sample counter that replaces the reference's global np.random state
(RobustART/noise/utils/imagenet_c/corruptions.py draws from np.random)."""

_M = 0xFFFFFFFF
_R = (13, 15, 26, 6, 17, 29, 16, 24)


ROUNDS = 13   # must match RART_THREEFRY_ROUNDS in csrc/rart_common.h


def threefry2x32(k0, k1, c0, c1, rounds=ROUNDS):
    ks = (k0 & _M, k1 & _M, (0x1BD11BDA ^ k0 ^ k1) & _M)
    x0 = (c0 + ks[0]) & _M
    x1 = (c1 + ks[1]) & _M
    for r in range(rounds):
        x0 = (x0 + x1) & _M
        rot = _R[r & 7]
        x1 = ((x1 << rot) | (x1 >> (32 - rot))) & _M
        x1 ^= x0
        if (r & 3) == 3:
            s = (r >> 2) + 1
            x0 = (x0 + ks[s % 3]) & _M
            x1 = (x1 + ks[(s + 1) % 3] + s) & _M
    return x0, x1


def ctr0(block_index, stream_id):
    return (block_index & 0x0FFFFFFF) | (stream_id << 28)


def host_uniform(seed, sample, stream_id, index=0):
    """U[0,1) double from 53 bits of one Threefry call; stream ids 8..15 are reserved for host draws."""
    w0, w1 = threefry2x32(seed & _M, (seed >> 32) & _M, ctr0(index, stream_id), sample & _M)
    return ((w0 >> 5) * 67108864.0 + (w1 >> 6)) / 9007199254740992.0


def host_uniform_many(seed, samples, stream_id, index=0):
    """host_uniform for an array of sample indices at once (numpy uint64 lanes; same values as the scalar form)."""
    import numpy as np
    M = np.uint64(_M)
    k0, k1 = np.uint64(seed & _M), np.uint64((seed >> 32) & _M)
    ks = (k0, k1, np.uint64(0x1BD11BDA) ^ k0 ^ k1)
    x0 = (np.full(len(samples), ctr0(index, stream_id), dtype=np.uint64) + ks[0]) & M
    x1 = ((np.asarray(samples, dtype=np.uint64) & M) + ks[1]) & M
    for r in range(ROUNDS):
        x0 = (x0 + x1) & M
        rot = np.uint64(_R[r & 7])
        x1 = ((x1 << rot) | (x1 >> (np.uint64(32) - rot))) & M
        x1 ^= x0
        if (r & 3) == 3:
            sft = (r >> 2) + 1
            x0 = (x0 + ks[sft % 3]) & M
            x1 = (x1 + ks[(sft + 1) % 3] + np.uint64(sft)) & M
    return ((x0 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (x1 >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


class _State:
    seed = 0
    sample_offset = 0


def manual_seed(seed, sample_offset=0):
    """Set the noise seed and reset the global sample counter.  Sample i of the k-th call gets
    global index sample_offset + (images already processed) + i, so results do not depend on how
    a dataset is batched or sharded (pass sample_offset = rank's first index under DDP)."""
    _State.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _State.sample_offset = int(sample_offset)


def next_offset(n):
    off = _State.sample_offset
    _State.sample_offset += int(n)
    return off


def current_seed():
    return _State.seed
