// device-less stand-ins for the two symbols the parser's translation units need from the GPU half (tools/dbg/parse_prof.sh)
#include <hip/hip_runtime_api.h>
extern "C" hipError_t hipGetDeviceCount(int *n) { *n = 0; return hipSuccess; }
extern "C" const char *hipGetErrorString(hipError_t) { return "stub"; }
extern "C" int ea1_tx_size_index(int wl, int hl)
{
    if (wl < 2 || wl > 6 || hl < 2 || hl > 6) return -1;
    const int d = wl - hl;   // squares 0-4, 2:1 shapes 5-12, 4:1 shapes 13-18 (av1_dev.h tx_size_index)
    if (d == 0) return wl - 2;
    if (d == -1) return 5 + 2 * (wl - 2);
    if (d == 1) return 6 + 2 * (hl - 2);
    if (d == -2) return 13 + 2 * (wl - 2);
    if (d == 2) return 14 + 2 * (hl - 2);
    return -1;
}
