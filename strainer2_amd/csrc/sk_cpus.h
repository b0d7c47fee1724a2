/* sk_cpus.h -- how many CPUs the host layer may count on: the online CPUs, cut down to the cgroup's CPU quota
 * where there is one (a container on a 256-CPU host may be granted 16 CPUs' worth of time; threads beyond that
 * only take turns), shared out among the processes of a one-process-per-GPU start on this node
 * (LOCAL_WORLD_SIZE, as torchrun sets it, or OMPI_COMM_WORLD_LOCAL_SIZE). */
#ifndef SK_CPUS_H
#define SK_CPUS_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static long sk_cpu_budget(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (n < 1) n = 1;
    if (f) {
        char q[32];
        long quota, period;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0 && sscanf(q, "%ld", &quota) == 1 && quota > 0) {
            const long share = quota / period < 1 ? 1 : quota / period;
            if (share < n) n = share;
        }
        fclose(f);
    }
    {
        const char *lw = getenv("LOCAL_WORLD_SIZE") ? getenv("LOCAL_WORLD_SIZE") : getenv("OMPI_COMM_WORLD_LOCAL_SIZE");
        const long ranks = lw ? atol(lw) : 1;
        if (ranks > 1) n = n / ranks < 1 ? 1 : n / ranks;
    }
    return n;
}

#endif
