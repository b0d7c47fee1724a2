// The partitioned pipeline (an experiment kept selectable: option "pipeline" = 2) -- part of sk_device.hip (included there, in this order; not a translation unit of its own).

// ---------------------------------------------------------------------------------------------
// The partitioned pipeline for large batches (the "LDS-staged probe"): the level-1 question "is this chunk's 16-mer in the
// strain at all?" costs one L2 request per chunk in sk_scan_grid, and the L2's request rate (about 270 G/s over its 128
// channels), not HBM, is what bounds that kernel.  Here the question is answered from LDS instead:
//   sk_bin        reads the stream once (the only full pass), decodes every chunk, hashes its canonical 16-mer and deals
//                 (chunk index in the tile, 20 hash bits) to one of SK_BIN_P partitions by 7 more hash bits: a fixed-size
//                 segment per (partition, tile) in HBM, 4 bytes per clean chunk = a quarter of the stream's bytes.
//   sk_lds_probe  one workgroup per partition (and share of the tiles) keeps that partition's slice of the filter
//                 -- a plain bitmap over the 20 bits, 128 KiB -- in LDS and streams the partition's segments past it;
//                 survivors (the strain's chunks plus a few per cent) set their byte in a per-chunk candidate map.
//   sk_scan_grid<.., CAND>  as before, but reading only the lines around candidates and asking the L2 filter only
//                 about them.
// Exactness is untouched: the filters only prune (no false negatives), stage 2 verifies what is left.
// ---------------------------------------------------------------------------------------------
#define SK_BIN_TILE   65536u
#define SK_BIN_CH     (SK_BIN_TILE / 16u)          // 4096 chunks: 12 bits
#define SK_BIN_P      128u                         // partitions: 7 bits
#define SK_BIN_CAP    40u                          // entries per (partition, tile): mean 28.5 clean chunks, +2 sigma; the rest go straight to the candidates
#define SK_BIN_WORDS  32768u                       // 2^20 bits per partition slice

// partition (7 bits) and in-partition key (20 bits) of a canonical 16-mer's mix; multiplier of its own, so that the slices'
// false positives are not the L2 filter's
__device__ __forceinline__ uint32_t sk_grid3_hash(uint32_t g) { return (g ^ (g >> 13)) * 0x5BD1E995u; }

__global__ void sk_grid3_insert(const sk_u4 *__restrict__ slots, uint64_t nslots, uint32_t *__restrict__ w3)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (from the resident table: built on first use only)
    if (i >= nslots) return;
    const uint64_t k = sk_slot_key(slots[i]);
    if (k == SK_EMPTY64) return;
    for (int off = 0; off < 16; off++) {
        const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
        const uint32_t r = sk_revcomp16(f);
        const uint32_t h = sk_grid3_hash(sk_gmix(f < r ? f : r));
        const uint32_t key = (h >> 5) & 0xFFFFFu;
        uint32_t *w = w3 + (size_t)(h >> 25) * SK_BIN_WORDS + (key >> 5);
        const uint32_t bit = 1u << (key & 31u);
        if (!(__builtin_nontemporal_load(w) & bit)) atomicOr(w, bit);
    }
}

__global__ __launch_bounds__(256)
void sk_bin(const uint8_t *__restrict__ stream, uint64_t nbytes, sk_table_view table, uint32_t *__restrict__ bins,
            uint8_t *__restrict__ bin_n, uint32_t ntiles, uint8_t *__restrict__ cand, uint32_t *__restrict__ flags)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[SK_BIN_P * SK_BIN_CAP];
    __shared__ uint32_t cnt[SK_BIN_P];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint64_t tile0 = (uint64_t)tile * SK_BIN_TILE;
    if (tid < SK_BIN_P) cnt[tid] = 0u;
    for (uint32_t i = tid; i < SK_BIN_P * SK_BIN_CAP; i += 256u) stage[i] = 0xFFFFFFFFu;     // "no entry"
    __syncthreads();
    uint32_t bad = 0;
    const bool inside = tile0 + SK_BIN_TILE <= nbytes;            // (workgroup-uniform)
#pragma unroll 1
    for (uint32_t half = 0; half < 2u; half++) {                  // 2 x 8 loads in flight per thread
        sk_u4 vv[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t c = tid + (half * 8u + (uint32_t)it) * 256u;
            vv[it] = inside ? sk_stream_load((const sk_u4 *)(stream + tile0 + (uint64_t)c * 16u))
                            : sk_load_chunk(stream, nbytes, (int64_t)(tile0 + (uint64_t)c * 16u));
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t c = tid + (half * 8u + (uint32_t)it) * 256u;
            if (tile0 + (uint64_t)c * 16u >= nbytes) continue;
            const sk_u4 v = vv[it];
            uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
            sk_decode4(v.x, c0, i0);
            sk_decode4(v.y, c1, i1);
            sk_decode4(v.z, c2, i2);
            sk_decode4(v.w, c3, i3);
            const uint32_t inv16 = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
            if (inv16) {
                if (sk_chunk_has_odd_byte(v, inv16)) {            // a byte for the byte-string kernel: note the chunk
                    bad = 1;
                    const uint32_t at = atomicAdd(&flags[2], 1u);
                    if (at < table.oddcap) table.oddlist[at] = (uint32_t)((tile0 >> 4) + c);
                }
                continue;
            }
            const uint32_t cw = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
            const uint32_t rc = sk_revcomp32(cw);
            const uint32_t h = sk_grid3_hash(sk_gmix(cw < rc ? cw : rc));
            const uint32_t p = h >> 25;
            const uint32_t ent = (c << 20) | ((h >> 5) & 0xFFFFFu);
            const uint32_t r = ent != 0xFFFFFFFFu ? atomicAdd(&cnt[p], 1u) : SK_BIN_CAP;
            if (r < SK_BIN_CAP) stage[p * SK_BIN_CAP + r] = ent;
            else cand[(tile0 >> 4) + c] = 1u;                     // no room in the segment: a candidate without being asked
        }
    }
    __syncthreads();
    // whole segments go out, 16 bytes per lane (SK_BIN_CAP is a multiple of 4): unused slots say "no entry"
    for (uint32_t i = tid; i < SK_BIN_P * SK_BIN_CAP / 4u; i += 256u) {
        const uint32_t p = i / (SK_BIN_CAP / 4u), r4 = i - p * (SK_BIN_CAP / 4u);
        ((sk_u4 *)(bins + ((size_t)p * ntiles + tile) * SK_BIN_CAP))[r4] = ((const sk_u4 *)stage)[i];
    }
    (void)bin_n;
    if (bad) atomicAdd(&flags[0], 1u);
}

__global__ __launch_bounds__(1024)
void sk_lds_probe(const uint32_t *__restrict__ w3, const uint32_t *__restrict__ bins, const uint8_t *__restrict__ bin_n,
                  uint32_t ntiles, uint32_t splits, uint8_t *__restrict__ cand)
{
    extern __shared__ uint32_t slice[];                           // SK_BIN_WORDS words = 128 KiB
    const uint32_t p = blockIdx.x / splits, sp = blockIdx.x % splits, tid = threadIdx.x;
    const sk_u4 *src = (const sk_u4 *)(w3 + (size_t)p * SK_BIN_WORDS);
    for (uint32_t i = tid; i < SK_BIN_WORDS / 4u; i += 1024u) ((sk_u4 *)slice)[i] = src[i];
    __syncthreads();
    const uint32_t t_lo = (uint32_t)((uint64_t)ntiles * sp / splits), t_hi = (uint32_t)((uint64_t)ntiles * (sp + 1u) / splits);
    const sk_u4 *seg = (const sk_u4 *)(bins + ((size_t)p * ntiles + t_lo) * SK_BIN_CAP);
    (void)bin_n;
    const uint32_t total4 = (t_hi - t_lo) * (SK_BIN_CAP / 4u);    // 16-byte groups of four entries
    uint8_t *const cbase = cand + (size_t)t_lo * SK_BIN_CH;
    auto judge = [&](uint32_t ent, uint32_t t) {
        const uint32_t key = ent & 0xFFFFFu;
        if (ent != 0xFFFFFFFFu && ((slice[key >> 5] >> (key & 31u)) & 1u)) cbase[(size_t)t * SK_BIN_CH + (ent >> 20)] = 1u;
    };
    for (uint32_t i0 = 0; i0 < total4; i0 += 4096u) {
        sk_u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + tid + 1024u * (uint32_t)k;
            v[k] = (sk_u4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (i < total4) v[k] = __builtin_nontemporal_load(seg + i);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = (i0 + tid + 1024u * (uint32_t)k) / (SK_BIN_CAP / 4u);
            judge(v[k].x, t); judge(v[k].y, t); judge(v[k].z, t); judge(v[k].w, t);
        }
    }
}
