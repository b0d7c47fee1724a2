/* tools/probes/gz_blocks_dump.c -- EXPERIMENT (not product code), host half of tools/probes/gpu_inflate_probe.hip.
 * Decodes one single-member .gz file with the library's serial decoder (sk_gzfast.h) and writes, for its first NB
 * dynamic-Huffman blocks: where the block's SYMBOLS start (bit offset), how many bytes it decodes to, where those go in
 * the text, and the block's decode tables exactly as the serial decoder uses them.  With that the GPU side can run the
 * decode loop of every block independently -- the question being asked is only how fast a GPU runs that loop.
 *     gcc -O2 -o gz_blocks_dump tools/probes/gz_blocks_dump.c -lpthread ; ./gz_blocks_dump reads.fq.gz out.bin [NB]
 * out.bin: u64 nblocks, u64 text_bytes_covered; nblocks x {u64 sym_bit, u64 out_off, u32 out_len, u32 pad};
 *          nblocks x {u32 litlen[6624], u32 dist[4096]} */
#define _GNU_SOURCE
#include <stdio.h>
#include "../../strainer2_amd/csrc/sk_gzfast.h"

static int sink(void *u, const unsigned char *d, size_t n) { (void)d; *(size_t *)u += n; return 0; }
typedef struct { uint64_t sym_bit, out_off; uint32_t out_len, pad; } blk;

int main(int argc, char **argv)
{
    struct stat st;
    unsigned char *m;
    skz_stream s;
    skz_tables *dyn = (skz_tables *)malloc(sizeof *dyn), *fixed = (skz_tables *)malloc(sizeof *fixed);
    size_t got = 0, nb = 0, cap = argc > 3 ? (size_t)atol(argv[3]) : 8192;
    blk *b = (blk *)calloc(cap + 1, sizeof *b);
    uint32_t *tabs;
    FILE *f;
    int fd, final = 0;
    if (argc < 3 || (fd = open(argv[1], O_RDONLY)) < 0 || fstat(fd, &st)) { fprintf(stderr, "usage: gz_blocks_dump file.gz out.bin [blocks]\n"); return 2; }
    m = (unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED || !skz_header(m, (size_t)st.st_size)) return 2;
    tabs = (uint32_t *)malloc(cap * (6624 + 4096) * sizeof(uint32_t));
    pthread_once(&skz_crc_once, skz_crc_init);
    skz_fixed_tables(fixed);
    memset(&s, 0, sizeof s);
    s.out_base = (unsigned char *)malloc(SKZ_WINDOW + SKZ_OUT_CHUNK + 1024);
    s.out_end = s.out_base + SKZ_WINDOW + SKZ_OUT_CHUNK;
    s.out = s.out_flushed = s.out_base;
    s.sink = sink; s.user = &got; s.skip_crc = 1;
    s.data = m; s.in = m + skz_header(m, (size_t)st.st_size); s.in_end = m + st.st_size;
    while (!final && nb < cap) {
        uint32_t type;
        uint64_t before;
        SKZ_REFILL(&s);
        final = (int)SKZ_BITS(&s, 1); SKZ_DROP(&s, 1);
        type = SKZ_BITS(&s, 2); SKZ_DROP(&s, 2);
        if (type != 2) { fprintf(stderr, "block %zu is of type %u: this probe wants dynamic blocks only\n", nb, type); break; }
        if (skz_read_dynamic(&s, dyn, 0)) { fprintf(stderr, "bad header\n"); return 1; }
        skz_flush(&s, 0);
        before = s.total;
        b[nb].sym_bit = skz_bit_position(&s);
        b[nb].out_off = before;
        if (skz_block(&s, dyn)) { fprintf(stderr, "bad block\n"); return 1; }
        skz_flush(&s, 0);
        b[nb].out_len = (uint32_t)(s.total - before);
        memcpy(tabs + nb * (6624 + 4096), dyn->litlen, 6624 * 4);
        memcpy(tabs + nb * (6624 + 4096) + 6624, dyn->dist, 4096 * 4);
        nb++;
    }
    f = fopen(argv[2], "wb");
    { uint64_t h[2] = {nb, s.total}; fwrite(h, 8, 2, f); }
    fwrite(b, sizeof *b, nb, f);
    fwrite(tabs, (6624 + 4096) * 4, nb, f);
    fclose(f);
    fprintf(stderr, "%zu blocks, %llu bytes of text, %.1f KB of text per block\n", nb, (unsigned long long)s.total, (double)s.total / (double)(nb ? nb : 1) / 1e3);
    return 0;
}
