"""Statistics of the dropout mask generator (stamp_amd/csrc/common.h: drop_rowkey / drop_pair_bits), restated in numpy:  python tools/dropout_hash_stats.py

For keep rates p in {0.25, 0.1, 0.5} and two row ranges, 4096 rows x 4096 elements each: deviation of the keep rate, covariances between neighbouring elements,
the two halves of a pair, elements 2 / 4 / 64 apart, rows 1 / 2 / 8 / 1025 apart and the diagonal -- each in units of its standard error under a fair Bernoulli
source -- and the spread of the row / column means relative to the binomial one.  |z| <= ~3 everywhere is what a fair source gives on this many statistics."""
import numpy as np

M = np.uint32
np.seterr(over="ignore")
GOLD = M(0x9E3779B1)


def fmix32(h):
    h = h.copy(); h ^= h >> M(16); h *= M(0x85EBCA6B); h ^= h >> M(13); h *= M(0xC2B2AE35); h ^= h >> M(16)
    return h


def drop_mix(h):
    h = h.copy(); h ^= h >> M(16); h *= M(0xC2B2AE35); h ^= h >> M(16)
    return h


def rowkey(seed, stream, row):
    a = fmix32(M(seed & 0xFFFFFFFF) ^ row.astype(M) ^ M((stream * 0x9E3779B1) & 0xFFFFFFFF))
    return fmix32(a + M(seed >> 32))


def main():
    worst = 0.0
    for base in (0, 1 << 20):
        rows = np.arange(base, base + 4096, dtype=np.uint64)
        rk = rowkey(0x123456789ABCDEF, 11, rows)
        pairs = np.arange(2048, dtype=M)
        y = drop_mix(rk[:, None] ^ (pairs[None, :] * GOLD))
        for p in (0.25, 0.1, 0.5):
            thr, q = M(round(p * 65536)), 1 - p
            m = np.empty((len(rows), 2 * len(pairs)), np.float64)
            m[:, 0::2] = (y & M(0xFFFF)) >= thr
            m[:, 1::2] = (y >> M(16)) >= thr
            c = m - q
            st = dict(mean=m.mean() - q, adj=(c[:, :-1] * c[:, 1:]).mean(), pair_halves=(c[:, 0::2] * c[:, 1::2]).mean(), k2=(c[:, :-2] * c[:, 2:]).mean(),
                      k4=(c[:, :-4] * c[:, 4:]).mean(), k64=(c[:, :-64] * c[:, 64:]).mean(), r1=(c[:-1] * c[1:]).mean(), r2=(c[:-2] * c[2:]).mean(),
                      r8=(c[:-8] * c[8:]).mean(), r1025=(c[:-1025] * c[1025:]).mean(), diag=(c[:-1, :-1] * c[1:, 1:]).mean())
            se = {k: (np.sqrt(p * q / m.size) if k == "mean" else p * q / np.sqrt(m.size)) for k in st}
            z = {k: v / se[k] for k, v in st.items()}
            rsd, csd = m.mean(1).std() / np.sqrt(p * q / m.shape[1]), m.mean(0).std() / np.sqrt(p * q / m.shape[0])
            worst = max(worst, max(abs(v) for v in z.values()))
            print(f"rows {base}.. p={p}: " + " ".join(f"{k}={v:+.1f}" for k, v in z.items()) + f"  row-mean sd x{rsd:.3f} column-mean sd x{csd:.3f}")
    print(f"largest |z| = {worst:.1f}")


if __name__ == "__main__":
    main()
