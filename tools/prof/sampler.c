/* A sampling profiler for the host side of a frame, for boxes without perf: LD_PRELOAD this, set WRPROF_OUT=<file>; a SIGPROF timer
 * (CPU time of the whole process, 4 kHz) records the interrupted PC of whichever thread took the signal; at exit the PCs and
 * /proc/self/maps go to the file, tools/prof/report.py resolves them with addr2line.
 *   gcc -O2 -fPIC -shared -o tools/prof/libsampler.so tools/prof/sampler.c */
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <stdint.h>
#define MAXS (1 << 20)
static uint64_t* pcs; static volatile long n;
static void on_prof(int sig, siginfo_t* si, void* uc_) {
  (void)sig; (void)si;
  ucontext_t* uc = (ucontext_t*)uc_;
  long i = __atomic_fetch_add(&n, 1, __ATOMIC_RELAXED);
  if (i < MAXS) pcs[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
}
static void dump(void) {
  const char* out = getenv("WRPROF_OUT");
  if (!out) return;
  struct itimerval z; memset(&z, 0, sizeof z); setitimer(ITIMER_PROF, &z, NULL);
  FILE* f = fopen(out, "w");
  if (!f) return;
  FILE* m = fopen("/proc/self/maps", "r");
  char line[1024];
  while (m && fgets(line, sizeof line, m)) if (strstr(line, " r-xp ") || strstr(line, " r-x")) fprintf(f, "M %s", line);
  if (m) fclose(m);
  long k = n < MAXS ? n : MAXS;
  for (long i = 0; i < k; i++) fprintf(f, "S %llx\n", (unsigned long long)pcs[i]);
  fclose(f);
}
__attribute__((constructor)) static void init(void) {
  if (!getenv("WRPROF_OUT")) return;
  pcs = (uint64_t*)calloc(MAXS, 8);
  struct sigaction sa; memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  struct itimerval t; t.it_interval.tv_sec = 0; t.it_interval.tv_usec = 250; t.it_value = t.it_interval;
  setitimer(ITIMER_PROF, &t, NULL);
  atexit(dump);
}
