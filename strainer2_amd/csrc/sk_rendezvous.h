/* sk_rendezvous.h -- how the ranks of one launch (one process per GPU) find each other before RCCL is up (internal).
 *
 * New relative to the reference, which is one process.  Rank 0 has to hand a 128-byte RCCL unique id to
 * the others; the only thing the processes share for sure is a file system path.  Three things can go wrong with "rank 0
 * writes a file, the others read it", and a blocking ncclCommInitRank turns each of them into a hang:
 *   - a file LEFT BEHIND by a run that crashed is read as if it were this launch's      -> every rank proves freshness
 *   - a rank that FAILED during its own set-up (strain unreadable, no device) has left  -> set-up status is part of the
 *     while the others walk into the collective                                            exchange; anyone failed = all leave
 *   - a rank that never shows up                                                        -> every wait is bounded
 *
 * Protocol (files <base>.hello.<r> for r = 1..n-1 and <base> itself, all written as tmp + rename):
 *   rank r > 0  draws a random 64-bit token, (re)writes hello.r = {magic, r, n, token, status} every 200 ms and polls <base>
 *               until it holds {magic, n, verdict, tokens[], payload} with tokens[r] == its token: a file from any other
 *               launch cannot know the token.
 *   rank 0      removes <base> and every hello.* it finds (leftovers; a live rank rewrites its hello within 200 ms), then
 *               polls until all n-1 hellos are there, and publishes <base> with everyone's token, the verdict
 *               (0 = go, 1 = somebody's set-up failed) and the payload.
 *               Just before it publishes it reads every hello ONCE MORE: a rank that gave up meanwhile (its own timeout) has
 *               removed its hello, and the others must not walk into a collective that is one rank short -- verdict 1.
 *   Every file carries the launch's NONCE (a hash of SK_LAUNCH_ID, TORCHELASTIC_RUN_ID or MASTER_ADDR:MASTER_PORT, whichever
 *   is set): two launches of one user that share the default path do not take each other's hellos or boards.
 *   Everyone returns SKR_OK / SKR_ABORT (verdict 1: leave without touching the collective) / SKR_TIMEOUT.
 * The path defaults to node-local /tmp: for several nodes SK_RCCL_ID_FILE must name a path all of them see.
 */
#ifndef SK_RENDEZVOUS_H
#define SK_RENDEZVOUS_H
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define SKR_OK       0
#define SKR_ABORT    1      /* some rank reported a failed set-up: nobody enters the collective */
#define SKR_TIMEOUT  2
#define SKR_IO       3
#define SKR_MAGIC    0x534B5244565A3032ull     /* "SKRDVZ02" */
#define SKR_MAX_WORLD 64
#define SKR_PAYLOAD  128

typedef struct { uint64_t magic; uint32_t rank, world; uint64_t token; uint32_t status, pad; uint64_t nonce; } skr_hello;
typedef struct { uint64_t magic; uint32_t world, verdict; uint64_t nonce; uint64_t tokens[SKR_MAX_WORLD]; unsigned char payload[SKR_PAYLOAD]; } skr_board;

/* what tells this launch from another one of the same user: whatever the launcher set for all its ranks */
static uint64_t skr_nonce(void)
{
    const char *names[] = {"SK_LAUNCH_ID", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"};
    uint64_t h = 0xCBF29CE484222325ull;
    int i, any = 0;
    for (i = 0; i < 4; i++) {
        const char *v = getenv(names[i]);
        if (!v || (i >= 2 && any == 1)) continue;          /* (an explicit launch id makes the address irrelevant) */
        if (i < 2) any = 1; else any = 2;
        for (; *v; v++) { h ^= (unsigned char)*v; h *= 0x100000001B3ull; }
        h ^= 0xFFu; h *= 0x100000001B3ull;
    }
    return any ? h : 0;
}

static double skr_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static uint64_t skr_token(void)
{
    uint64_t t = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&t, sizeof t, 1, f) != 1) t = 0; fclose(f); }
    if (!t) { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); t = ((uint64_t)ts.tv_nsec << 32) ^ (uint64_t)ts.tv_sec ^ ((uint64_t)getpid() << 17); }
    return t | 1u;                                   /* never 0 */
}

static int skr_write_atomic(const char *path, const void *data, size_t n)
{
    char tmp[600];
    FILE *f;
    snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
    f = fopen(tmp, "wb");
    if (!f) return -1;
    if (fwrite(data, 1, n, f) != n) { fclose(f); unlink(tmp); return -1; }
    fclose(f);
    if (rename(tmp, path) != 0) { unlink(tmp); return -1; }
    return 0;
}

static int skr_read_whole(const char *path, void *data, size_t n)
{
    FILE *f = fopen(path, "rb");
    size_t got;
    if (!f) return -1;
    got = fread(data, 1, n, f);
    fclose(f);
    return got == n ? 0 : -1;
}

/* my_status: 0 = this rank's set-up went fine, else it failed.  payload: in on rank 0, out on the others (128 bytes). */
static int skr_exchange(int rank, int world, const char *base, int my_status, unsigned char payload[SKR_PAYLOAD], double timeout_s)
{
    char path[560];
    const double t0 = skr_now();
    int r;
    if (world < 1 || world > SKR_MAX_WORLD || rank < 0 || rank >= world || !base) return SKR_IO;
    if (world == 1) return my_status ? SKR_ABORT : SKR_OK;
    if (rank == 0) {
        skr_board bd;
        uint64_t seen = 1;                           /* bit r: hello of rank r seen (world <= 64) */
        memset(&bd, 0, sizeof bd);
        unlink(base);
        for (r = 1; r < world; r++) { snprintf(path, sizeof path, "%s.hello.%d", base, r); unlink(path); }
        bd.magic = SKR_MAGIC; bd.world = (uint32_t)world; bd.verdict = my_status ? 1u : 0u; bd.nonce = skr_nonce();
        for (;;) {
            for (r = 1; r < world; r++) {
                skr_hello h;
                if ((seen >> r) & 1u) continue;
                snprintf(path, sizeof path, "%s.hello.%d", base, r);
                if (skr_read_whole(path, &h, sizeof h) == 0 && h.magic == SKR_MAGIC && h.rank == (uint32_t)r && h.world == (uint32_t)world && h.token &&
                    h.nonce == bd.nonce) {
                    bd.tokens[r] = h.token;
                    if (h.status) bd.verdict = 1u;
                    seen |= (uint64_t)1 << r;
                }
            }
            if (seen == (world == 64 ? ~(uint64_t)0 : (((uint64_t)1 << world) - 1))) break;
            if (skr_now() - t0 > timeout_s) {
                /* tell whoever did arrive to leave, then give up */
                bd.verdict = 1u;
                memcpy(bd.payload, payload, SKR_PAYLOAD);
                skr_write_atomic(base, &bd, sizeof bd);
                return SKR_TIMEOUT;
            }
            usleep(20000);
        }
        /* everybody was seen -- but is everybody still there?  A rank that ran into its own timeout while this one waited
         * for a later rank has taken its hello away: publishing "go" now would send the others into a collective that is
         * one rank short */
        for (r = 1; r < world; r++) {
            skr_hello h;
            snprintf(path, sizeof path, "%s.hello.%d", base, r);
            if (skr_read_whole(path, &h, sizeof h) != 0 || h.magic != SKR_MAGIC || h.token != bd.tokens[r] || h.nonce != bd.nonce) bd.verdict = 1u;
        }
        memcpy(bd.payload, payload, SKR_PAYLOAD);
        if (skr_write_atomic(base, &bd, sizeof bd) != 0) return SKR_IO;
        return bd.verdict ? SKR_ABORT : SKR_OK;
    } else {
        skr_hello h;
        skr_board bd;
        double last_write = -1.0;
        memset(&h, 0, sizeof h);
        h.magic = SKR_MAGIC; h.rank = (uint32_t)rank; h.world = (uint32_t)world; h.token = skr_token(); h.status = my_status ? 1u : 0u; h.nonce = skr_nonce();
        snprintf(path, sizeof path, "%s.hello.%d", base, rank);
        for (;;) {
            const double now = skr_now();
            if (now - last_write > 0.2) {            /* rank 0 may have swept it away as a leftover: say it again */
                if (skr_write_atomic(path, &h, sizeof h) != 0) return SKR_IO;
                last_write = now;
            }
            if (skr_read_whole(base, &bd, sizeof bd) == 0 && bd.magic == SKR_MAGIC && bd.world == (uint32_t)world && bd.nonce == h.nonce && bd.tokens[rank] == h.token) {
                memcpy(payload, bd.payload, SKR_PAYLOAD);
                unlink(path);
                return bd.verdict ? SKR_ABORT : SKR_OK;
            }
            if (now - t0 > timeout_s) { unlink(path); return SKR_TIMEOUT; }
            usleep(20000);
        }
    }
}

#endif
