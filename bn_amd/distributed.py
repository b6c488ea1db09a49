"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The pairing path shards by independent units:
  * pairing_batch / g*_mul_batch: contiguous index ranges per rank, NO data-path collective.
  * pairing_product (multi-pairing): each rank reduces its shard to ONE un-exponentiated Fq12 (384 bytes), one
    all_gather of 48 x u64 per rank (RCCL has no user-defined reduction; 3 KiB on 8 GPUs, latency-bound, bandwidth
    irrelevant), then every rank multiplies the `world` partials and runs a SINGLE final exponentiation.  Equal to the
    reference fold prod_i pairing(p_i, q_i) (shootout/main.rs:11-16) bit for bit, because the final exponentiation is a
    homomorphism, Fq12 multiplication is commutative and every output is canonical.

`TorchEngine` adapts bn_amd.Engine's device-pointer API to torch tensors (int64 storage of the u64 limbs).
Anything with the same five methods can be injected (the gloo tests inject an oracle-backed double; the product never does).
"""
import numpy as np

GT_WORDS = 48


def shard_range(n, rank, world):
    """contiguous [lo, hi) of rank `rank` (same rule as the C driver would use: lo = n*rank/world)"""
    return n * rank // world, n * (rank + 1) // world


class TorchEngine:
    """bn_amd.Engine on torch CUDA tensors (dtype int64 viewing the reference's u64 limbs)"""

    def __init__(self, engine, device):
        import torch
        self.torch = torch
        self.e = engine
        self.device = device

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.int64, device=self.device)

    def pairing_batch(self, p, q, out=None):
        n = p.shape[0]
        out = self.empty(n, GT_WORDS) if out is None else out
        self.e.pairing_batch_dev(p.data_ptr(), q.data_ptr(), out.data_ptr(), n, self._stream())
        return out

    def miller_product(self, p, q):
        out = self.empty(GT_WORDS)
        self.e.miller_product_dev(p.data_ptr(), q.data_ptr(), p.shape[0], out.data_ptr(), self._stream())
        return out

    def g2_prepare(self, q):
        """native tables of this rank's G2 points (bn254_g2_prepare_dev): a PreparedG2 handle on this rank's GPU"""
        return self.e.g2_prepare_dev(q.data_ptr(), q.shape[0], self._stream())

    def miller_product_prepared(self, p, prepared):
        out = self.empty(GT_WORDS)
        self.e.miller_product_prepared_native_dev(p.data_ptr(), prepared, p.shape[0], out.data_ptr(), stream=self._stream())
        return out

    def gt_product(self, vals):
        out = self.empty(GT_WORDS)
        vals = vals.contiguous()
        self.e.gt_product_dev(vals.data_ptr(), vals.shape[0], out.data_ptr(), self._stream())
        return out

    def product_final_exp(self, vals):
        """final_exponentiation(prod vals): ONE launch for the tail of the sharded multi-pairing"""
        out = self.empty(GT_WORDS)
        vals = vals.contiguous()
        self.e.gt_product_final_exp_dev(vals.data_ptr(), vals.shape[0], out.data_ptr(), self._stream())
        return out

    def final_exp(self, f):
        out = self.empty(GT_WORDS)
        self.e.final_exp_batch_dev(f.data_ptr(), out.data_ptr(), 1, self._stream())
        return out

    def g1_mul(self, p, k, normalize=True):
        out = self.torch.empty_like(p)
        self.e.g1_mul_dev(p.data_ptr(), k.data_ptr(), out.data_ptr(), p.shape[0], self._stream(), normalize)
        return out

    def g2_mul(self, p, k, normalize=True):
        out = self.torch.empty_like(p)
        self.e.g2_mul_dev(p.data_ptr(), k.data_ptr(), out.data_ptr(), p.shape[0], self._stream(), normalize)
        return out

    def gt_pow(self, a, k):
        out = self.torch.empty_like(a)
        self.e.gt_pow_dev(a.data_ptr(), k.data_ptr(), out.data_ptr(), a.shape[0], self._stream())
        return out


def all_gather_partials(partial, group=None):
    """the ONE exchange step of the multi-pairing: (48,) int64 per rank -> (world, 48)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):          # no process group: a single process, nothing to exchange
        return partial.reshape(1, GT_WORDS)                          # (an initialised group of ONE rank still goes through the collective)
    world = dist.get_world_size(group)
    src = partial.contiguous().reshape(GT_WORDS)
    if dist.get_backend(group) != "nccl" and src.is_cuda:          # gloo (tests, one-GPU smoke runs): exchange through the host
        out = torch.empty(world * GT_WORDS, dtype=src.dtype)
        dist.all_gather_into_tensor(out, src.cpu(), group=group)
        return out.reshape(world, GT_WORDS).to(partial.device)
    out = torch.empty(world * GT_WORDS, dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.reshape(world, GT_WORDS)


def pairing_product_sharded(eng, p_local, q_local, group=None):
    """multi-pairing product over all ranks' shards; every rank returns the same Gt (48 words)"""
    partial = eng.miller_product(p_local, q_local)          # local: Miller loops + Fq12 product tree
    parts = all_gather_partials(partial, group)             # RCCL all-gather of 384 B per rank
    if hasattr(eng, "product_final_exp"):
        return eng.product_final_exp(parts)                 # world-1 multiplications + ONE final exponentiation, one launch
    return eng.final_exp(eng.gt_product(parts))


def pairing_product_prepared_sharded(eng, p_local, prepared_local, group=None):
    """the same over natively prepared points: `prepared_local` = eng.g2_prepare(this rank's shard of the G2 points), made once and re-used;
    the exchange and the tail are those of pairing_product_sharded"""
    partial = eng.miller_product_prepared(p_local, prepared_local)
    parts = all_gather_partials(partial, group)
    if hasattr(eng, "product_final_exp"):
        return eng.product_final_exp(parts)
    return eng.final_exp(eng.gt_product(parts))


def pairing_batch_sharded(eng, p_local, q_local, out=None):
    """independent pairings of this rank's shard: no collective"""
    return eng.pairing_batch(p_local, q_local, out)


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs: scalars from SplitMix64, points r*G1 / s*G2 built ON THE DEVICE by the
# reference's own double-and-add chain so that z != 1, as `G::random` (groups/mod.rs:220-222) gives in benches/api.rs:156-160
SEED = 0x424E323534            # "BN254"
_U = 4965661367192848881
R_MOD = 36 * _U**4 + 36 * _U**3 + 18 * _U**2 + 6 * _U + 1
Q_MOD = 36 * _U**4 + 36 * _U**3 + 24 * _U**2 + 6 * _U + 1
_M64 = (1 << 64) - 1


def _splitmix64_words(stream_ids, nwords):
    """nwords successive SplitMix64 outputs for every stream id (vectorised); stream s starts at state SEED + s*2^32"""
    state = (np.uint64(SEED) + (stream_ids.astype(np.uint64) << np.uint64(32)))
    out = np.empty((stream_ids.shape[0], nwords), np.uint64)
    with np.errstate(over="ignore"):
        for w in range(nwords):
            state = state + np.uint64(0x9E3779B97F4A7C15)
            z = state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            out[:, w] = z ^ (z >> np.uint64(31))
    return out


def synthetic_scalars(lo, hi, which):
    """Fr scalars for pairing indices [lo,hi): a 512-bit draw reduced mod r (distribution of arith.rs:195-198), returned as
    the reference's Montgomery limbs (n,4) uint64.  which = 0 -> G1 scalars (stream 2i), 1 -> G2 scalars (stream 2i+1)."""
    idx = np.arange(lo, hi, dtype=np.uint64) * np.uint64(2) + np.uint64(which)
    words = _splitmix64_words(idx, 8)
    out = np.empty((hi - lo, 4), np.uint64)
    for j in range(hi - lo):
        v = 0
        for w in range(8):
            v |= int(words[j, w]) << (64 * w)
        m = v % R_MOD * (1 << 256) % R_MOD
        out[j] = [(m >> (64 * i)) & _M64 for i in range(4)]
    return out


def generator_limbs():
    """(G1::one() 12 words, G2::one() 24 words) as numpy uint64 (groups/mod.rs:355-361, 377-390)"""
    def mont(v): return [(v * (1 << 256) % Q_MOD >> (64 * i)) & _M64 for i in range(4)]
    g1 = mont(1) + mont(2) + mont(1)
    g2x = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634)
    g2y = (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531)
    g2 = mont(g2x[0]) + mont(g2x[1]) + mont(g2y[0]) + mont(g2y[1]) + mont(1) + [0, 0, 0, 0]
    return np.array(g1, np.uint64), np.array(g2, np.uint64)


def synthetic_scalars_device(eng, lo, hi, which):
    """the same scalars as synthetic_scalars(lo, hi, which), generated in HBM by bn254_synthetic_scalars_dev ((n,4) int64 tensor);
    word-for-word equality with the numpy version is a GPU test"""
    out = eng.empty(hi - lo, 4)
    eng.e.synthetic_scalars_dev(SEED, lo, hi - lo, which, out.data_ptr(), eng._stream())
    return out


def synthetic_points(eng, lo, hi):
    """device tensors (P (n,12), Q (n,24)) for indices [lo,hi): r_i * G1::one(), s_i * G2::one() by the reference's own
    double-and-add chain (Jacobian, z != 1), scalars and generator tiles produced on the device"""
    torch = eng.torch
    n = hi - lo
    g1, g2 = generator_limbs()
    k1 = synthetic_scalars_device(eng, lo, hi, 0)
    k2 = synthetic_scalars_device(eng, lo, hi, 1)
    t1 = torch.from_numpy(g1.view(np.int64)).to(eng.device); t2 = torch.from_numpy(g2.view(np.int64)).to(eng.device)
    b1 = eng.empty(n, 12); b2 = eng.empty(n, 24)
    eng.e.tile_dev(t1.data_ptr(), 96, n, b1.data_ptr(), eng._stream())
    eng.e.tile_dev(t2.data_ptr(), 192, n, b2.data_ptr(), eng._stream())
    P = eng.g1_mul(b1, k1, normalize=False)
    Q = eng.g2_mul(b2, k2, normalize=False)
    torch.cuda.synchronize(eng.device)
    return P, Q
