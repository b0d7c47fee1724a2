/* tools/probes/sigprof_preload.c -- a sampling profiler for boxes without perf: LD_PRELOAD this, and the process is sampled
 * on its CPU time (ITIMER_PROF, SK_PROF_HZ per second, default 1000): program counter + the running thread's name.  At exit
 * (the program must RETURN from main: SK_LEAK_AT_EXIT=0 for the programs here) every sample is written to SK_PROF_OUT
 * (default sigprof.txt) as "thread-name module-path offset"; tools/sigprof_report.py turns that into a table by thread
 * class and function with addr2line, on any machine that holds the same binaries.
 *   gcc -O2 -shared -fPIC -o /tmp/sigprof.so tools/probes/sigprof_preload.c -ldl
 *   SK_LEAK_AT_EXIT=0 LD_PRELOAD=/tmp/sigprof.so strainer2_amd/bin/strain_detect ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>

#define CAP (1u << 20)
typedef struct { uintptr_t pc; char name[16]; } sample;
static sample *buf;
static volatile unsigned nsamp;

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
    ucontext_t *uc = (ucontext_t *)uc_;
    const unsigned i = __atomic_fetch_add(&nsamp, 1u, __ATOMIC_RELAXED);
    (void)sig; (void)si;
    if (i >= CAP) return;
    buf[i].pc = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
    prctl(PR_GET_NAME, buf[i].name, 0, 0, 0);
}

typedef struct { uintptr_t pc; const char *path; uintptr_t base; } find;
static int phdr_cb(struct dl_phdr_info *info, size_t size, void *data)
{
    find *f = (find *)data;
    int j;
    (void)size;
    for (j = 0; j < info->dlpi_phnum; j++)
        if (info->dlpi_phdr[j].p_type == PT_LOAD) {
            const uintptr_t a = info->dlpi_addr + info->dlpi_phdr[j].p_vaddr;
            if (f->pc >= a && f->pc < a + info->dlpi_phdr[j].p_memsz) { f->path = info->dlpi_name; f->base = info->dlpi_addr; return 1; }
        }
    return 0;
}

__attribute__((constructor)) static void prof_start(void)
{
    struct sigaction sa;
    struct itimerval it;
    const int hz = getenv("SK_PROF_HZ") ? atoi(getenv("SK_PROF_HZ")) : 1000;
    buf = (sample *)calloc(CAP, sizeof *buf);
    if (!buf || hz <= 0) return;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000000 / hz;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((destructor)) static void prof_dump(void)
{
    struct itimerval off;
    FILE *f;
    unsigned i, n;
    char exe[512];
    ssize_t el = readlink("/proc/self/exe", exe, sizeof exe - 1);
    exe[el > 0 ? el : 0] = 0;
    memset(&off, 0, sizeof off);
    setitimer(ITIMER_PROF, &off, NULL);
    n = nsamp < CAP ? nsamp : CAP;
    f = fopen(getenv("SK_PROF_OUT") ? getenv("SK_PROF_OUT") : "sigprof.txt", "w");
    if (!f) return;
    for (i = 0; i < n; i++) {
        find fd;
        fd.pc = buf[i].pc; fd.path = NULL; fd.base = 0;
        dl_iterate_phdr(phdr_cb, &fd);
        buf[i].name[15] = 0;
        fprintf(f, "%s\t%s\t%lx\n", buf[i].name[0] ? buf[i].name : "?", fd.path && fd.path[0] ? fd.path : (fd.path ? exe : "[unknown]"), (unsigned long)(buf[i].pc - fd.base));
    }
    fclose(f);
}
