// cpu_scale_probe -- how many host cores does this box really give one process?  N threads each run the same fixed amount of dependent
// integer work (a 64-bit LCG with a data-dependent branch: nothing the memory system can limit); the aggregate rate against N shows where
// the scaling ends -- the cgroup's CPU quota, the SMT pairs, or the machine.  Prints the affinity mask's size and cgroup cpu.max beside it.
//   g++ -O2 -std=c++17 -pthread tools/cpu_scale_probe.cpp -o tools/bin/cpu_scale_probe && tools/bin/cpu_scale_probe
#include <sched.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static uint64_t spin(uint64_t seed, uint64_t iters)
{
    uint64_t x = seed, acc = 0;
    for (uint64_t i = 0; i < iters; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        if (x >> 63) acc += x >> 32;
        else acc ^= x;
    }
    return acc;
}

int main()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) printf("affinity mask: %d cpus; hardware_concurrency %u\n", CPU_COUNT(&set), std::thread::hardware_concurrency());
    for (const char *p : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"}) {
        std::ifstream f(p);
        std::string line;
        if (f && std::getline(f, line)) printf("%s: %s\n", p, line.c_str());
    }
    const uint64_t iters = 200000000ull;
    double base = 0;
    for (int n : {1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128}) {
        std::vector<std::thread> th;
        std::vector<uint64_t> out(size_t(n), 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; k++) th.emplace_back([&, k] { out[size_t(k)] = spin(uint64_t(k) + 1, iters); });
        for (auto &t : th) t.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t sink = 0;
        for (uint64_t v : out) sink ^= v;
        const double rate = double(n) * double(iters) / s / 1e9;
        if (n == 1) base = rate;
        printf("%3d threads: %7.3f s, %8.2f G iterations/s aggregate = %6.2f x one thread  (%llx)\n", n, s, rate, rate / base, (unsigned long long)(sink & 0xF));
        fflush(stdout);
    }
    return 0;
}
