"""Seeded input generators and strict comparers shared by the parity tests.

The generators restate the reference's test fixtures with fixed seeds (the reference draws
from std::random_device): tests/test_utils.cpp:293-350 (contiguous wrapper) and :695-773
(paged wrapper with a shuffled page pool).  Data follow the reference distribution --
floats U(0,1]*ratio (src/kernels/rand_assign.cu:13), ints U[0,max] (:23) -- or, with
conditioned=True, U(-1,1)/sqrt(D)-scaled weights so that softmax is not one-hot.
"""
import numpy as np

PAGE = 16
TOL = 1e-3  # absolute; the reference's default threshold (include/kernels/utils.cuh:27-39)


def rand_f(rng, shape, ratio=1.0):
    # curand_uniform is (0, 1]; 1 - U[0,1) reproduces the half-open side
    return ((1.0 - rng.random(shape, dtype=np.float32)) * np.float32(ratio)).astype(np.float32)


def rand_i(rng, shape, max_val):
    return rng.integers(0, max_val + 1, size=shape, dtype=np.int64).astype(np.int32)


def naive_case(seed, n_batch, n_sequence, input_dim, output_dim, conditioned=False, zero_every=None,
               lengths=None):
    """tests/test_utils.cpp:293-341 generate_device_and_host_tensors, seeded."""
    rng = np.random.default_rng(seed)
    n_new = int(rng.integers(1, n_batch + 1))
    new_idx = rand_i(rng, (n_batch,), n_batch - 1)
    new_idx[:n_new] = rng.permutation(n_batch)[:n_new].astype(np.int32)
    c = {}
    if conditioned:
        c["inp"] = (rng.random((n_batch, n_sequence, input_dim), dtype=np.float32) * 2 - 1).astype(np.float32)
        sc = np.float32(1.0 / np.sqrt(input_dim))
        for w in ("wk", "wq", "wv"):
            c[w] = ((rng.random((input_dim, output_dim), dtype=np.float32) * 2 - 1) * sc * 2).astype(np.float32)
    else:
        c["inp"] = rand_f(rng, (n_batch, n_sequence, input_dim))
        for w in ("wk", "wq", "wv"):
            c[w] = rand_f(rng, (input_dim, output_dim))
    c["lengths"] = rand_i(rng, (n_batch,), n_sequence) if lengths is None else np.asarray(lengths, np.int32).copy()
    if zero_every:
        c["lengths"][::zero_every] = 0
    c["new_batch_idx"] = new_idx
    c["n_new"] = n_new
    # outputs start from random contents: "regions the op must not touch stay untouched" is part of the contract
    c["kt_cache"] = rand_f(rng, (n_batch, output_dim, n_sequence))
    c["v_cache"] = rand_f(rng, (n_batch, n_sequence, output_dim))
    c["q_output"] = rand_f(rng, (n_batch, output_dim))
    c["qkt_output"] = rand_f(rng, (n_batch, n_sequence))
    c["attention_result"] = rand_f(rng, (n_batch, output_dim))
    return c


def build_page_pool(rng, lengths, n_sequence, emb_dim, spare_blocks=0):
    """tests/test_utils.cpp:703-734: ceil(min(L+1,S)/16) pages per non-empty row from a shuffled pool.
    Returns (pool float32 [n_blocks * 48 * D] filled with random data, table int64 [B, S/16] of float offsets, -1 = none)."""
    B = len(lengths)
    width = n_sequence // PAGE
    per_row = [0 if L == 0 else -(-min(int(L) + 1, n_sequence) // PAGE) for L in lengths]
    total = int(sum(per_row)) + spare_blocks
    block = PAGE * 3 * emb_dim
    pool = rand_f(rng, (max(total, 1) * block,))
    order = rng.permutation(max(total, 1))
    table = np.full((B, width), -1, np.int64)
    cur = 0
    for b in range(B):
        for j in range(min(per_row[b], width)):
            table[b, j] = int(order[cur]) * block
            cur += 1
    return pool, table


def paged_case(seed, n_batch, n_sequence, emb_dim, conditioned=False, zero_every=None, lengths=None):
    """tests/test_utils.cpp:695-773 generate_paged_attention_wrapper_device_tensors, seeded (host side)."""
    rng = np.random.default_rng(seed)
    c = naive_case(seed + 1, n_batch, n_sequence, emb_dim, emb_dim, conditioned=conditioned)
    c["lengths"] = rand_i(rng, (n_batch,), n_sequence - 1) if lengths is None else np.asarray(lengths, np.int32).copy()
    if zero_every:
        c["lengths"][::zero_every] = 0
    c["inp_embedding"] = c.pop("inp")
    if conditioned:
        c["kt_cache"] = (c["kt_cache"] * 2 - 1).astype(np.float32)
        c["v_cache"] = (c["v_cache"] * 2 - 1).astype(np.float32)
        c["q_output"] = (c["q_output"] * 2 - 1).astype(np.float32)
    c["pool"], c["table"] = build_page_pool(rng, c["lengths"], n_sequence, emb_dim)
    return c


def assert_close(actual, expected, thr=TOL, what=""):
    """Stricter than the reference's device comparer (src/kernels/utils.cu:24,37): NaN/Inf on either side fails."""
    a = np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape, (what, a.shape, e.shape)
    assert np.isfinite(a).all(), f"{what}: non-finite values in actual"
    assert np.isfinite(e).all(), f"{what}: non-finite values in expected"
    diff = np.abs(a.astype(np.float64) - e.astype(np.float64))
    worst = float(diff.max()) if diff.size else 0.0
    assert worst <= thr, f"{what}: max |diff| = {worst:.3e} > {thr:g} at {np.unravel_index(diff.argmax(), diff.shape)}"


def assert_close_rel(actual, expected, rel=2e-6, abs_=TOL, what=""):
    """For raw scores whose magnitude makes 1e-3 absolute meaningless in fp32 (|x| ~ 1e5): 1e-3 + rel*|x|."""
    a = np.asarray(actual, np.float64)
    e = np.asarray(expected, np.float64)
    assert np.isfinite(a).all() and np.isfinite(e).all(), what
    bound = abs_ + rel * np.abs(e)
    bad = np.abs(a - e) > bound
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, worst {np.abs(a - e).max():.3e}"


def well_posed_rows(raw_scores, lengths, min_gap=6.0):
    """Rows whose softmax is a well-posed comparison target at 1e-3 absolute.  On the reference's U(0,1] data the raw
    scores are ~1e4..1e5, so two fp32 implementations differ by O(1e-2) in a score DIFFERENCE; a probability moves by
    p * that, so a row qualifies when every runner-up is far enough below the top score that its probability (and its
    error) is negligible: gap >= min_gap means runner-up probabilities <= e^-6 ~ 2.5e-3 and errors ~1e-4.  Rows of length
    0 / 1 always qualify.  (The reference compares these probabilities at 1e-3 with curand data it never fixes,
    tests/paged_attention_kernels_test.cpp:114-233; near-ties make that comparison ill-posed, not wrong.)"""
    s = np.asarray(raw_scores, np.float64)
    ok = np.zeros(len(lengths), bool)
    for b, L in enumerate(lengths):
        L = int(L)
        if L <= 1:
            ok[b] = True
            continue
        top2 = np.partition(s[b, :L], L - 2)[L - 2:]
        ok[b] = (top2[1] - top2[0]) >= min_gap
    return ok


def assert_equal(actual, expected, what=""):
    a = np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape and a.dtype == e.dtype, (what, a.shape, e.shape, a.dtype, e.dtype)
    bad = a != e
    assert not bad.any(), f"{what}: {int(bad.sum())} mismatches, first at {np.argwhere(bad)[0]}"


def pool_index(table, b, s, seg, emb_dim):
    """Float offset of element (b, s, seg, 0) in the pool -- the layout rule of include/utils.h:32-60."""
    return table[b, s // PAGE] + (s % PAGE) * 3 * emb_dim + seg * emb_dim


def scatter_rows_to_pool(pool, table, rows_bs, seg, values):
    """pool[(b, s, seg, :)] = values[i] for (b, s) = rows_bs[i]."""
    if len(rows_bs) == 0:
        return
    D = values.shape[1]
    b = np.asarray([r[0] for r in rows_bs])
    s = np.asarray([r[1] for r in rows_bs])
    base = table[b, s // PAGE] + (s % PAGE) * 3 * D + seg * D
    assert (table[b, s // PAGE] >= 0).all()
    pool[(base[:, None] + np.arange(D)[None, :]).ravel()] = values.ravel()


def gather_rows_from_pool(pool, table, rows_bs, seg, emb_dim):
    b = np.asarray([r[0] for r in rows_bs])
    s = np.asarray([r[1] for r in rows_bs])
    base = table[b, s // PAGE] + (s % PAGE) * 3 * emb_dim + seg * emb_dim
    return pool[(base[:, None] + np.arange(emb_dim)[None, :])]


def bf16_bits(x):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


_FP8_TABLE = None


def _fp8_table():
    """The 127 non-negative finite OCP e4m3fn values, indexed by their byte code 0x00 .. 0x7e (0x7f is NaN):
    e = 0: m * 2^-9 (subnormal), else (1 + m / 8) * 2^(e - 7); largest 448."""
    global _FP8_TABLE
    if _FP8_TABLE is None:
        codes = np.arange(127)
        e, m = codes >> 3, codes & 7
        _FP8_TABLE = np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7)).astype(np.float64)
    return _FP8_TABLE


def fp8_bits_by_search(x):
    """float32 -> OCP e4m3fn byte codes (uint8), the definition spelled out: the nearest representable value, ties to the even
    code, SATURATING at +-448 (the format has no infinity; the kernels clamp before v_cvt_pk_fp8_f32, which would give NaN from
    465 on), NaN -> 0x7f.  Slow (float64 + a table search): kept as the model fp8_bits is checked against."""
    x = np.ascontiguousarray(x, np.float32)
    t = _fp8_table()
    a = np.minimum(np.abs(x.astype(np.float64)), 448.0)
    hi = np.clip(np.searchsorted(t, a, side="left"), 0, 126)          # first code with value >= a
    lo = np.maximum(hi - 1, 0)
    d_hi, d_lo = t[hi] - a, a - t[lo]
    code = np.where(d_hi < d_lo, hi, np.where(d_lo < d_hi, lo, np.where(hi % 2 == 0, hi, lo)))
    code = np.where(np.isnan(x), 0x7f, code).astype(np.uint8)
    return (code | (np.signbit(x).astype(np.uint8) << 7)).astype(np.uint8)


def fp8_bits(x):
    """The same codes by integer arithmetic on the float32 bits (what makes the fp8 tests' pools affordable): normal range --
    round the 23-bit mantissa to 3 bits, nearest even, carry into the exponent, rebias 127 -> 7; below 2^-6 -- the subnormal
    grid, rint(|x| * 2^9) (8 is the first normal code).  tests/test_oracle.py checks it against fp8_bits_by_search."""
    x = np.ascontiguousarray(x, np.float32)
    a = np.minimum(np.abs(x), np.float32(448.0))
    u = a.view(np.uint32)
    normal = ((u + np.uint32(0x7FFFF) + ((u >> np.uint32(20)) & np.uint32(1))) >> np.uint32(20)).astype(np.int32) - (120 << 3)
    with np.errstate(invalid="ignore"):
        sub = np.rint(np.nan_to_num(a) * np.float32(512.0)).astype(np.int32)
    code = np.where(a < np.float32(2.0 ** -6), sub, normal)
    code = np.where(np.isnan(x), 0x7f, code).astype(np.uint8)
    return (code | (np.signbit(x).astype(np.uint8) << 7)).astype(np.uint8)


_FP8_DECODE = None


def fp8_decode(bits):
    """OCP e4m3fn byte codes -> float32 (a 256-entry table)."""
    global _FP8_DECODE
    if _FP8_DECODE is None:
        mag = np.concatenate([_fp8_table(), [np.nan]]).astype(np.float32)
        _FP8_DECODE = np.concatenate([mag, -mag]).astype(np.float32)
    return _FP8_DECODE[np.asarray(bits, np.uint8)]


def fp8_round(x):
    """float32 values rounded to the nearest OCP e4m3fn value (saturating), returned as float32."""
    return fp8_decode(fp8_bits(x)).reshape(np.shape(x))


def bf16_round(x):
    """float32 values rounded to the nearest bfloat16, returned as float32."""
    return (bf16_bits(x).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(x))
