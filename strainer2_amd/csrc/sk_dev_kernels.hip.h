// The byte-string kernel and the kernels around the table: load, difference array, gather, build on the device, filter inserts -- part of sk_device.hip (included there, in this order; not a translation unit of its own).

// ---------------------------------------------------------------------------------------------
// wide (byte-string) path
// ---------------------------------------------------------------------------------------------
__constant__ signed char sk_comp_dev[256];

struct sk_wide_view {
    const char     *keys31;     // [nwide][32]
    const uint32_t *rows;       // [nwide]
    const uint32_t *index;      // [wmask+1]  0 = empty, else key index + 1
    uint32_t        wmask;
    uint32_t        nwide;
};

template <bool TALLY, bool UNION = false>
__global__ __launch_bounds__(256)
void sk_scan_wide(const uint8_t *__restrict__ stream, uint64_t nbytes, uint64_t emit_begin,
                  sk_table_view table, sk_wide_view wide, sk_sink sink,
                  const uint32_t *__restrict__ flags, uint32_t *__restrict__ next_flags)
{
    // the NEXT launch's flag words (the context alternates between two sets) are zeroed here, behind this launch's scan kernel and
    // before the next one's: a memset per scan less on the stream
    if (blockIdx.x == 0u && threadIdx.x < 4u) next_flags[threadIdx.x] = 0u;
    if (flags[0] == 0u) return;                        // no window with a non-ACGT byte in this batch
    // Work list: phase 1 of the scan kernel noted every 16-byte chunk that holds such a byte (flags[2] of them).
    // A window that needs this kernel contains one; it is handled from the chunk that holds its LAST non-ACGT
    // byte, so every window is handled once and the cost follows the number of odd bytes, not the batch size.
    // If the list overflowed, every position of the batch is visited instead.
    const uint32_t nodd = flags[2];
    const bool listed = nodd <= table.oddcap;
    const uint64_t nitems = listed ? (uint64_t)nodd * 46u : nbytes;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; item < nitems; item += stride) {
        uint64_t p, owner = ~0ull;
        if (listed) { owner = table.oddlist[item / 46u]; p = owner * 16u + item % 46u; }
        else p = item;
        if (p < emit_begin || p >= nbytes || p < (uint64_t)(SK_K - 1)) continue;
        const uint8_t *w = stream + (p - (SK_K - 1));
        char u[SK_K];
        bool hard = false, pure = true;
        int last_odd = -1;
        for (int i = 0; i < SK_K; i++) {
            const uint32_t b = w[i];
            hard |= (bool)sk_is_hard_break(b);
            if (!sk_is_acgt(b)) { pure = false; last_odd = i; }
            u[i] = (char)sk_upper(b);
        }
        if (hard || pure) continue;                    // skipped by the reference / done by the scan kernel
        if (listed && ((p - (SK_K - 1) + (uint64_t)last_odd) >> 4) != owner) continue;   // another chunk's window
        // orientation: sign of (window - revcomp) in signed-char order (src/genome_compare.c:1122-1141)
        int sign = 0;
        for (int i = 0; i < SK_K && sign == 0; i++) {
            const signed char f = (signed char)u[i];
            const signed char r = sk_comp_dev[(uint8_t)u[SK_K - 1 - i]];
            sign = (f > r) - (r > f);
        }
        char o[SK_K + 1];
        if (sign >= 0) { for (int i = 0; i < SK_K; i++) o[i] = u[i]; }
        else           { for (int i = 0; i < SK_K; i++) o[SK_K - 1 - i] = (char)sk_comp_dev[(uint8_t)u[i]]; }
        o[SK_K] = 0;
        bool opure = true, has_n = false, has_nul = false;
        for (int i = 0; i < SK_K; i++) {
            opure &= (bool)sk_is_acgt((uint8_t)o[i]) & ((uint8_t)o[i] < 'a');
            has_n |= (o[i] == 'N');
            has_nul |= (o[i] == 0);
        }
        if (has_n || has_nul) continue;
        if (opure) {                                   // e.g. a window with U whose revcomp wins
            uint64_t key = 0;
            for (int i = 0; i < SK_K; i++) key = (key << 2) | sk_code((uint8_t)o[i]);
            sk_probe<TALLY, false, UNION>(key, table, sink, (uint32_t)p);
        } else if (wide.nwide) {
            uint32_t slot = sk_hash_wide(o) & wide.wmask;
            for (;;) {
                const uint32_t e = wide.index[slot];
                if (e == 0u) break;
                const char *cand = wide.keys31 + (size_t)(e - 1u) * 32u;
                bool same = true;
                for (int i = 0; i < SK_K; i++) same &= (cand[i] == o[i]);
                if (same) { sk_on_hit<TALLY, false, UNION>(sink, wide.rows[e - 1u], (uint32_t)p); break; }
                slot = (slot + 1u) & wide.wmask;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// table build
// ---------------------------------------------------------------------------------------------
__global__ void sk_fill64(uint64_t *p, uint64_t n, uint64_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void sk_table_insert(const uint64_t *__restrict__ in, uint32_t n, sk_u4 *slots, uint32_t mask, uint32_t *flags,
                                const uint32_t *__restrict__ perm, const uint32_t *__restrict__ locality)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    const uint32_t idx = perm ? perm[i] : i;
    if (k == SK_EMPTY64) return;                       // wide row: not in this table
    if (k > SK_KMASK62) { atomicAdd(&flags[1], 1u); return; }
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot],
                                                (unsigned long long)SK_EMPTY64, (unsigned long long)k);
        if (old == SK_EMPTY64) {
            ((uint32_t *)&slots[slot])[2] = idx;
            ((uint32_t *)&slots[slot])[3] = 0xFFFFFFFEu | (locality ? locality[i] >> 31 : 0u);   // no text position (yet)
            return;
        }
        if (old == k) { atomicAdd(&flags[1], 1u); return; }     // duplicate key
        slot = (slot + 1u) & mask;
    }
}

// TALLY: the records with at least one hit, as {record, all hits, informative hits} (unordered); *n counts them
__global__ void sk_tally_compact(const uint32_t *__restrict__ tally, uint32_t nrec, uint32_t *__restrict__ out, unsigned long long *n)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint2 t = r < nrec ? ((const uint2 *)tally)[r] : make_uint2(0u, 0u);
    const unsigned long long m = __ballot(t.x != 0u);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(m);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(n, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), (int)leader) << 32) | (uint32_t)__shfl((int)(uint32_t)base, (int)leader);
    if (t.x != 0u) {
        uint32_t *o = out + 3ull * (base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull)));
        o[0] = r; o[1] = t.x; o[2] = t.y;
    }
}

// TALLY: one bit per counter index, set where the type column holds `value`
__global__ void sk_inf_bitmap(const uint32_t *__restrict__ type, uint32_t n, uint32_t value, uint32_t *__restrict__ bits)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool is = i < n && type[i] == value;
    const unsigned long long m = __ballot(is);
    if ((threadIdx.x & 63u) == 0u && i < n + 64u) { bits[2u * (i >> 6)] = (uint32_t)m; bits[2u * (i >> 6) + 1u] = (uint32_t)(m >> 32); }
}

// text position of every row into its table slot (pos_by_idx: by counter index, 0xFFFFFFFF = none)
__global__ void sk_table_setpos(sk_u4 *slots, uint64_t nslots, const uint32_t *__restrict__ pos_by_idx)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = slots[i];
        if (sk_slot_key(e) == SK_EMPTY64) continue;
        ((uint32_t *)&slots[i])[3] = (pos_by_idx[e.z] << 1) | (e.w & 1u);
    }
}

// ---- the difference array of the column being scanned, folded into it: counts[i] += diff[0] + ... + diff[i] ----
#define SK_DIFF_PER_BLOCK 4096u                        // 256 threads x 16 entries
__global__ __launch_bounds__(256) void sk_diff_block_sums(const uint32_t *__restrict__ diff, uint32_t n, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * SK_DIFF_PER_BLOCK + threadIdx.x * 16u;
    uint32_t t = 0;
    for (uint32_t i = 0; i < 16u; i++) if (base + i < n) t += diff[base + i];
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 128u; d > 0u; d >>= 1) { if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(1024) void sk_diff_scan_sums(uint32_t *sums, uint32_t nb)     // one block: exclusive scan in place
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023u) / 1024u, lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t i = lo; i < lo + per && i < nb; i++) t += sums[i];
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - t;              // exclusive prefix of this thread's slice
    for (uint32_t i = lo; i < lo + per && i < nb; i++) { const uint32_t v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void sk_diff_apply(uint32_t *__restrict__ diff, uint32_t n, const uint32_t *__restrict__ sums,
                                                      uint32_t *__restrict__ counts, uint32_t nrows)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * SK_DIFF_PER_BLOCK + threadIdx.x * 16u;
    uint32_t v[16], t = 0;
    for (uint32_t i = 0; i < 16u; i++) { v[i] = base + i < n ? diff[base + i] : 0u; t += v[i]; }
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 256u; d <<= 1) {
        const uint32_t u = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += u;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - t;
    for (uint32_t i = 0; i < 16u; i++) {
        run += v[i];
        if (base + i < nrows && run) counts[base + i] += run;
        if (base + i < n && v[i]) diff[base + i] = 0u;
    }
}

// counter columns live in "locality order" on the device (perm: caller's row -> counter index)
__global__ void sk_gather_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void sk_scatter_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}

// the permutation is the caller's locality[] without its orientation bit
__global__ void sk_perm_from_locality(uint32_t *__restrict__ perm, const uint32_t *__restrict__ locality, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = locality[i] & 0x7FFFFFFFu;
}

__global__ void sk_invert_perm(uint32_t *__restrict__ inv, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (perm[i] >= n) { atomicAdd(&flags[1], 1u); return; }
    inv[perm[i]] = i;
}

// grid filters: the canonical form of every 16-mer of every key, into both levels
__global__ void sk_grid_insert(const uint64_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ w1, uint32_t nblocks1,
                               uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    if (k == SK_EMPTY64) return;
    for (int off = 0; off < 16; off++) {
        const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
        const uint32_t r = sk_revcomp16(f);
        const uint32_t g = sk_gmix(f < r ? f : r);
        const uint32_t a = sk_grid1_bits(g);
        // consecutive keys share 15 of their 16 sub-words: most bits are set already, so look before the atomic
        uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
        const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
        if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
        if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
    }
    for (int off = 0; off < 8; off++) sk_grid2_insert24(w2, shift2, (k >> (2 * (7 - off))) & 0xFFFFFFFFFFFFull);   // level 2: its eight 24-mers
}

// ---- the table built ON THE DEVICE from the strain's 2-bit text (sk_table_build_from_text; strain_detect's opening) -----------------
// The host hands over the text (records end to end) and one bit per position "a window of 31 A/C/G/T bases of one record starts
// here" (src/genome_compare.c:1000-1019: every such window's oriented form is a key).  Rows are numbered by first occurrence along
// the text, which makes the counter index of a text position its RANK -- the layout the scan's verify stage wants anyway.
__device__ __forceinline__ bool sk_bit(const uint32_t *__restrict__ bits, uint32_t p) { return (bits[p >> 5] >> (p & 31u)) & 1u; }
__device__ __forceinline__ uint64_t sk_text_canon(const uint32_t *__restrict__ text2, uint32_t p, uint32_t *is_fwd)
{
    const uint64_t fwd = sk_text_key(text2, p);
    uint64_t r = ((uint64_t)__builtin_bitreverse32((uint32_t)fwd) << 32) | __builtin_bitreverse32((uint32_t)(fwd >> 32));
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    const uint64_t rc = (~r) >> 2;
    *is_fwd = fwd > rc;
    return fwd > rc ? fwd : rc;
}
// every window's key into the slots; the slot keeps the LOWEST position of its key (with the orientation it has there)
__global__ void sk_build_insert(const uint32_t *__restrict__ text2, const uint32_t *__restrict__ startok, uint32_t nbases, sk_u4 *slots, uint32_t mask)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases || !sk_bit(startok, p)) return;
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot], (unsigned long long)SK_EMPTY64, (unsigned long long)k);
        if (old == SK_EMPTY64 || old == k) { atomicMin(&((uint32_t *)&slots[slot])[3], (p << 1) | fw); return; }
        slot = (slot + 1u) & mask;
    }
}
__device__ __forceinline__ uint32_t sk_build_find(const sk_u4 *slots, uint32_t mask, uint64_t k)
{
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    while (sk_slot_key(slots[slot]) != k) slot = (slot + 1u) & mask;       // (the key is there: sk_build_insert put it)
    return slot;
}
// the positions at which a key occurs for the first time: the rank map's bits
__global__ void sk_build_first(const uint32_t *__restrict__ text2, const uint32_t *__restrict__ startok, uint32_t nbases, const sk_u4 *slots, uint32_t mask, sk_u4 *rank)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases || !sk_bit(startok, p)) return;
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    if ((slots[sk_build_find(slots, mask, k)].w >> 1) != p) return;
    atomicOr((uint32_t *)&rank[p >> 6] + ((p & 32u) ? 2 : 1), 1u << (p & 31u));          // (.y: bits 0..31 of the block, .z: 32..63)
}
// rank[b].x = first-occurrence positions before block b (one workgroup walks the blocks in slices; *total = all of them)
__global__ __launch_bounds__(1024) void sk_build_rank_scan(sk_u4 *rank, uint32_t nblk, uint32_t *total)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nblk + 1023u) / 1024u, lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) t += (uint32_t)__popc(rank[i].y) + (uint32_t)__popc(rank[i].z);
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - t;
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const uint32_t v = (uint32_t)__popc(rank[i].y) + (uint32_t)__popc(rank[i].z); rank[i].x = run; run += v; }
    if (threadIdx.x == 1023u) *total = part[1023];
}
// every row's counter index (= the rank of its first position) into its slot, and its key into the row-ordered key list
__global__ void sk_build_index(const uint32_t *__restrict__ text2, uint32_t nbases, sk_u4 *slots, uint32_t mask, const sk_u4 *__restrict__ rank, uint64_t *__restrict__ keys_by_row)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases) return;
    const sk_u4 r = rank[p >> 6];
    const uint64_t bits = ((uint64_t)r.z << 32) | r.y;
    if (!((bits >> (p & 63u)) & 1ull)) return;
    const uint32_t idx = r.x + (uint32_t)__popcll(bits & ((1ull << (p & 63u)) - 1ull));
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    ((uint32_t *)&slots[sk_build_find(slots, mask, k)])[2] = idx;
    keys_by_row[idx] = k;
}
__global__ void sk_set_rows_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ rows, uint32_t n, const uint32_t *__restrict__ perm, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm ? perm[rows[i]] : rows[i]] = v;
}
__global__ void sk_gather_keys(uint64_t *__restrict__ out, const uint64_t *__restrict__ keys_by_row, const uint32_t *__restrict__ rows, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keys_by_row[rows[i]];
}
__global__ void sk_fill32(uint32_t *p, uint32_t n, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the same set from the table's slots (a table loaded without a text stage: built when the first scan needs it)
__global__ void sk_grid_insert_slots(const sk_u4 *__restrict__ slots, uint64_t nslots, uint32_t *__restrict__ w1, uint32_t nblocks1,
                                     uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const uint64_t k = sk_slot_key(slots[i]);
        if (k == SK_EMPTY64) continue;
        for (int off = 0; off < 16; off++) {
            const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
            const uint32_t r = sk_revcomp16(f);
            const uint32_t g = sk_gmix(f < r ? f : r);
            const uint32_t a = sk_grid1_bits(g);
            uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
            const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
            if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
            if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
        }
        for (int off = 0; off < 8; off++) sk_grid2_insert24(w2, shift2, (k >> (2 * (7 - off))) & 0xFFFFFFFFFFFFull);
    }
}

// ... and from the strain's TEXT, when every key is a window of it: a key's sixteen 16-mers are the text's 16-mers at its
// place and the fifteen places behind it, and consecutive keys share fifteen of them -- inserting every 16-mer of the text ONCE
// does in nbases steps what the key-wise kernels do in 16 x nrows (5.5 ms -> 0.3 ms for a 5 Mbp strain; 167 ms -> 10 ms for the
// union of 32).  16-mers that no key holds (across an N, across two records) only add a few false positives.
__global__ void sk_grid_insert_text(const uint32_t *__restrict__ text2, uint32_t nbases, uint32_t *__restrict__ w1, uint32_t nblocks1,
                                    uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q + 16u > nbases) return;
    const uint32_t w = q >> 4, o2 = 2u * (q & 15u);
    const uint32_t f = (uint32_t)(((((uint64_t)text2[w] << 32) | text2[w + 1u]) << o2) >> 32);
    const uint32_t r = sk_revcomp16(f);
    const uint32_t g = sk_gmix(f < r ? f : r);
    const uint32_t a = sk_grid1_bits(g);
    uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
    const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
    if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
    if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
    if (q + 24u <= nbases) sk_grid2_insert24(w2, shift2, sk_text_24(text2, q));       // level 2: the text's 24-mer at this place
}
