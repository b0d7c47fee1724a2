// The union table's kernels (one table for several resident strains) -- part of sk_device.hip (included there, in this order; not a translation unit of its own).

// ---------------------------------------------------------------------------------------------
// One table for several resident strains (strain_detect -S, BASELINE configs[4]: 32 strains per GPU against one
// metagenome).  Without it a batch of reads is scanned once per strain; with it once.  New relative to the reference,
// which holds one strain per process (src/strain_detect.c:137-146).
//
// The union is built on the device from the members' resident tables: global row g = base[s] + the member's counter
// index; the members' texts and rank maps are laid one behind the other (each padded to a multiple of 64 bases), so
// that seed-and-verify works on the union as it does on one strain.  A key that several strains hold gets ONE slot (the
// first member's: its row and text position), but every one of its global rows carries the same pair of masks
// {members that hold the key, members in which it is informative}: whichever row a hit comes out at -- the slot's, or
// the rank of a position in another member's text -- it names the same strains.  The hit log names, per informative
// strain, that member's own row (sk_union_post looks the key up in the member's table).
// ---------------------------------------------------------------------------------------------
struct sk_union_member { const sk_u4 *slots; uint32_t mask; const uint32_t *inv; };

__global__ void sk_union_insert(const sk_u4 *__restrict__ mslots, uint64_t nslots, uint32_t base, uint32_t tbase,
                                sk_u4 *slots, uint32_t mask, uint64_t *__restrict__ ukeys)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = mslots[i];
        const uint64_t k = sk_slot_key(e);
        if (k == SK_EMPTY64) continue;
        const uint32_t g = base + e.z;
        ukeys[g] = k;
        const uint32_t tp = e.w >> 1;
        const uint32_t w = tp == 0x7FFFFFFFu ? e.w : (((tbase + tp) << 1) | (e.w & 1u));
        uint32_t slot = sk_slot0(sk_khash(k), mask);
        for (;;) {
            const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot], (unsigned long long)SK_EMPTY64, (unsigned long long)k);
            if (old == SK_EMPTY64) { ((uint32_t *)&slots[slot])[2] = g; ((uint32_t *)&slots[slot])[3] = w; break; }
            if (old == k) break;                           // an earlier member holds the key: its slot stands
            slot = (slot + 1u) & mask;
        }
    }
}

// pass A: every row finds its key's slot row ("canon") and sets its member's bits there
__global__ void sk_union_mask_a(const sk_u4 *__restrict__ mslots, uint64_t nslots, uint32_t base, uint32_t s,
                                const uint32_t *__restrict__ type, uint32_t inf_value,
                                const sk_u4 *__restrict__ slots, uint32_t mask, uint2 *umask, uint32_t *__restrict__ canon)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = mslots[i];
        const uint64_t k = sk_slot_key(e);
        if (k == SK_EMPTY64) continue;
        uint32_t slot = sk_slot0(sk_khash(k), mask), w = 0xFFFFFFFFu;
        for (;;) {
            const sk_u4 u = slots[slot];
            const uint64_t uk = sk_slot_key(u);
            if (uk == k) { w = u.z; break; }
            if (uk == SK_EMPTY64) break;                   // (cannot happen: every key was inserted)
            slot = (slot + 1u) & mask;
        }
        const uint32_t g = base + e.z;
        canon[g] = w == 0xFFFFFFFFu ? g : w;
        if (w == 0xFFFFFFFFu) continue;
        atomicOr(&umask[w].x, 1u << s);
        if (type[e.z] == inf_value) atomicOr(&umask[w].y, 1u << s);
    }
}

// pass B: the other rows of a key copy the pair
__global__ void sk_union_mask_b(uint2 *umask, const uint32_t *__restrict__ canon, uint32_t n)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t w = canon[g];
    if (w != g && w < n) umask[g] = umask[w];
}

// a member's rank map into the union's: counter indices move by the member's base
__global__ void sk_union_rank_copy(sk_u4 *__restrict__ dst, const sk_u4 *__restrict__ src, uint32_t nsrc, uint32_t ndst, uint32_t base, uint32_t after)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ndst) return;
    if (b < nsrc) { sk_u4 r = src[b]; r.x += base; dst[b] = r; }
    else dst[b] = (sk_u4){after, 0u, 0u, 0u};             // padding: no row starts here
}

// The scan logged (position, global row) once per informative hit.  Dealt out here: one entry per strain in which the row's key
// is informative, {position, strain << SK_UNION_ROW_BITS | that member's OWN row of the key} (a probe of the key in the member's
// table).  *nout counts every entry, stored or not (the caller asks again with more room if it exceeds cap_out).
struct sk_union_resolve_args {
    const uint2 *raw; const unsigned long long *nraw; unsigned long long cap_raw;
    uint2 *out; unsigned long long *nout; unsigned long long cap_out;
    const uint64_t *ukeys; const uint2 *umask; const sk_union_member *mem;
};
__device__ __forceinline__ void sk_union_resolve_block(const sk_union_resolve_args &a, uint32_t block, uint32_t nblocks)
{
    const uint2 *__restrict__ raw = a.raw; uint2 *__restrict__ out = a.out; unsigned long long *nout = a.nout;
    const unsigned long long cap_out = a.cap_out;
    const uint64_t *__restrict__ ukeys = a.ukeys; const uint2 *__restrict__ umask = a.umask; const sk_union_member *__restrict__ mem = a.mem;
    const unsigned long long n = *a.nraw < a.cap_raw ? *a.nraw : a.cap_raw;
    const unsigned long long stride = (unsigned long long)nblocks * blockDim.x;
    for (unsigned long long i = (unsigned long long)block * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint2 e = raw[i];
        uint32_t infm = umask[e.y].y;
        const uint64_t k = ukeys[e.y];
        unsigned long long at = atomicAdd(nout, (unsigned long long)__popc(infm));
        while (infm) {
            const uint32_t s = (uint32_t)__builtin_ctz(infm);
            infm &= infm - 1u;
            const sk_union_member m = mem[s];
            uint32_t slot = sk_slot0(sk_khash(k), m.mask), row = (1u << SK_UNION_ROW_BITS) - 1u;
            for (;;) {
                const sk_u4 u = m.slots[slot];
                const uint64_t uk = sk_slot_key(u);
                if (uk == k) { row = m.inv ? m.inv[u.z] : u.z; break; }
                if (uk == SK_EMPTY64) break;
                slot = (slot + 1u) & m.mask;
            }
            if (at < cap_out) out[at] = make_uint2(e.x, (s << SK_UNION_ROW_BITS) | row);
            at++;
        }
    }
}

// The (record, strain) tallies of a union scan live in a dense array that is ALL ZERO between launches: the scan marks the
// records it touched (one byte each), and this pass visits only those -- appends {record * ns + strain, all, informative} for
// the pairs that were hit, zeroes them again and clears the mark.  Its cost follows the reads that hit a strain, not
// records x strains (57 MB per 32 MiB batch of 150-base reads and 32 strains; zeroing and sweeping that much took longer
// than the scan itself).
#define SK_UC_SETS_MAX 16u                                          // at most this many sets of 64 records per wave (a workgroup of four waves: 4096 records)
__device__ __forceinline__ void sk_union_compact_block(uint2 *__restrict__ tally, uint8_t *__restrict__ flag, uint32_t nrec, uint32_t ns,
                                                       uint32_t *__restrict__ out, unsigned long long *n, uint32_t sets, uint32_t block)
{
    // A wave looks at 64 records' marks at once; the marked ones (about one in fifty) are then taken two at a time, half a wave
    // per record with one lane per strain: a row of 32 tallies is ONE 256-byte read and the non-zero ones find their places by
    // ballot and popcount.  The workgroup counts first, claims its room with ONE atomic, then writes: the room counter is one
    // address for the whole card, and the memory side does about 120 M returning atomics a second on one address -- with one
    // atomic per marked record (rounds 2-4) this kernel took 587 us for the 71 K pairs of a 512 MiB batch, 35 us for the 4 K of a
    // 32 MiB one, next to scans of 787 and 69 us (profiles/r04_union_launch_trace.txt).  `sets` trades the atomics (one per
    // workgroup) against the wave's chain of dependent reads (a set after the other): the host picks it for a few hundred workgroups.
    __shared__ uint32_t wcount[4];
    __shared__ unsigned long long wbase[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = lane >> 5, sl = lane & 31u;
    const uint32_t first = (block * 4u + wave) * (sets * 64u);                     // this wave's first record
    unsigned long long at = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t cnt = 0;
        for (uint32_t set = 0; set < sets; set++) {
            const uint32_t r0 = first + set * 64u, r = r0 + lane;
            if (r0 >= nrec) break;
            const bool marked = r < nrec && flag[r] != 0;
            uint64_t m = __ballot(marked);
            if (pass && marked) flag[r] = 0;
            while (m) {
                const uint32_t b0 = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint32_t b1 = 64u;
                if (m) { b1 = (uint32_t)__builtin_ctzll(m); m &= m - 1; }
                const uint32_t mine = half ? b1 : b0;
                for (uint32_t sb = 0; sb < ns; sb += 32u) {        // (SK_UNION_MAX is 32: one round)
                    const uint32_t st = sb + sl;
                    const bool act = mine < 64u && st < ns;
                    uint2 *const cell = tally + (size_t)(r0 + (act ? mine : 0u)) * ns + (act ? st : 0u);
                    uint2 t = make_uint2(0u, 0u);
                    if (act) t = *cell;
                    const bool nz = act && (t.x | t.y) != 0u;
                    const uint64_t bm = __ballot(nz);
                    if (pass && nz) {
                        const unsigned long long o = at + cnt + (unsigned long long)__popcll(bm & ((1ull << lane) - 1ull));
                        out[3 * o] = (r0 + mine) * ns + st; out[3 * o + 1] = t.x; out[3 * o + 2] = t.y;
                        *cell = make_uint2(0u, 0u);
                    }
                    cnt += (uint32_t)__popcll(bm);
                }
            }
        }
        if (pass) break;
        if (lane == 0u) wcount[wave] = cnt;
        __syncthreads();
        if (threadIdx.x == 0u) {
            const uint32_t total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
            unsigned long long base = total ? atomicAdd(n, (unsigned long long)total) : 0ull;
            for (uint32_t w = 0; w < 4u; w++) { wbase[w] = base; base += wcount[w]; }
        }
        __syncthreads();
        if (!cnt) break;                                           // (this wave has nothing to write; wave-uniform)
        at = wbase[wave];
    }
}

// Both of the above in one launch (they do not depend on each other, only on the scan): the first `ncompact` workgroups compact,
// the others deal the log out.
__global__ void __launch_bounds__(256) sk_union_post(uint2 *__restrict__ tally, uint8_t *__restrict__ flag, uint32_t nrec, uint32_t ns,
                                                     uint32_t *__restrict__ out, unsigned long long *n, uint32_t sets, uint32_t ncompact,
                                                     const sk_union_resolve_args ra)
{
    if (blockIdx.x < ncompact) sk_union_compact_block(tally, flag, nrec, ns, out, n, sets, blockIdx.x);
    else sk_union_resolve_block(ra, blockIdx.x - ncompact, gridDim.x - ncompact);
}

// The results' way home: the three counters, the first `eager` compacted pairs and log entries written straight into the page-locked
// landing area on the host (`land` is its device address), and the OTHER set of counters zeroed for the next launch.  One kernel
// where there were a memset and three copies, each its own trip through the runtime: 0.18 -> see profiles/r04_union_summary.txt.
__global__ void sk_union_ship(const unsigned long long *__restrict__ cnt, unsigned long long *__restrict__ cnt_next,
                              const uint32_t *__restrict__ compact, unsigned long long pairs_cap,
                              const uint2 *__restrict__ hits, unsigned long long hits_cap, uint32_t eager, uint8_t *__restrict__ land)
{
    const unsigned long long c0 = cnt[0], c1 = cnt[1], c2 = cnt[2];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    unsigned long long nr = c1 < pairs_cap ? c1 : pairs_cap, nh = c2 < hits_cap ? c2 : hits_cap;
    if (nr > eager) nr = eager;
    if (nh > eager) nh = eager;
    uint32_t *lr = (uint32_t *)(land + 64);
    uint2 *lh = (uint2 *)(land + 64 + (size_t)eager * 12u);
    for (unsigned long long i = t; i < 3ull * nr; i += nt) lr[i] = compact[i];
    for (unsigned long long i = t; i < nh; i += nt) lh[i] = hits[i];
    if (t == 0) {
        unsigned long long *lc = (unsigned long long *)land;
        lc[0] = c0; lc[1] = c1; lc[2] = c2;
        cnt_next[0] = 0ull; cnt_next[1] = 0ull; cnt_next[2] = 0ull; cnt_next[3] = 0ull;
    }
}
