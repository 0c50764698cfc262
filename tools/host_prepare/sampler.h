#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
#include <map>
static uint64_t g_samples[1 << 20];
static volatile int g_ns = 0;
static void prof_handler(int, siginfo_t *, void *uc) { if (g_ns < (1 << 20)) g_samples[g_ns++] = ((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP]; }
static void prof_start() { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = prof_handler; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, 0); itimerval it{{0, 200}, {0, 200}}; setitimer(ITIMER_PROF, &it, 0); }
static void prof_dump(const char *path) { itimerval it{{0, 0}, {0, 0}}; setitimer(ITIMER_PROF, &it, 0); std::map<uint64_t, int> h; for (int i = 0; i < g_ns; i++) h[g_samples[i]]++; FILE *f = fopen(path, "w"); for (auto &kv : h) fprintf(f, "%llx %d\n", (unsigned long long)kv.first, kv.second); fclose(f); fprintf(stderr, "%d samples\n", g_ns); }
