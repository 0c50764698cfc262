// Inverse-transform kernels (dav1d's itxfm_add family, src/itx_tmpl.c; AV1 spec 7.13.3).
//
// Work decomposition: a 64-lane workgroup owns 64 / max(min(H,32), W) transform blocks of one
// size; each lane runs whole 1-D transforms in registers (av1_itx_dev.h), rows then columns, with
// the coefficient tile and the intermediate tile in LDS (row stride padded to an odd number of
// dwords).  blockIdx.z selects the chain (one frame of every chain per launch).
// Inter blocks: residual added to the prediction in place.  Intra blocks: residual stored to the
// chain's int16 residual plane; the wavefront kernel adds it once the prediction exists.
// Traffic per block: w*h int16 coefficients read once, w*h pixels read-modify-written (or w*h
// int16 written) once, destination rows written by consecutive lanes.  Integer VALU, HBM/L2 bound.
#include "av1_dev.h"
#include "av1_itx_dev.h"
#include "ea1_common.h"

namespace ea1 {

using itx::TxLds;

template <int W, int H, int CLASS>
__global__ void __launch_bounds__(64)
itx_batch_kernel(Batch b)
{
    using T = TxLds<W, H>;
    constexpr int BPW = 64 / T::THREADS;
    __shared__ __attribute__((aligned(16))) int16_t lds[BPW * T::TOTAL];
    const FrameView *fvp = b.v[blockIdx.z];
    if (!fvp) return;
    const FrameView &fv = *fvp;
    const int count = fv.tx_count[CLASS];
    if (int(blockIdx.x) * BPW >= count) return;  // uniform
    const int g = threadIdx.x / T::THREADS, sub = threadIdx.x % T::THREADS;
    const int idx = blockIdx.x * BPW + g;
    const bool active = idx < count;
    const TxItem it = fv.tx_items[CLASS][active ? idx : 0];
    const int stride = fv.work.stride[it.plane];
    uint8_t *dst = fv.work.p[it.plane] + size_t(it.y) * stride + it.x;
    int16_t *resid = it.intra ? fv.resid[it.plane] + size_t(it.y) * stride + it.x : nullptr;
    // one wave per workgroup: groups taking different nz branches (each with its own barriers) is safe
    itx::tx_block_add_nz<W, H>(active ? it.nz_log2 : 5, lds + g * T::TOTAL, fv.coefs + it.coef_off, dst, stride, it.txtp, sub, active, resid, stride, active && it.corner);
}

template <int W, int H>
static void launch_class(const SubmitInfo &s, hipStream_t st, Prof *prof, const char *name)
{
    constexpr int CLASS = tx_size_index(itx::clog2(W), itx::clog2(H));
    static_assert(CLASS >= 0, "not an AV1 transform size");
    int max_count = 0;
    uint64_t bytes = 0;
    for (int c = 0; c < s.batch.n; c++) {
        if (!s.stats[c]) continue;
        max_count = std::max(max_count, s.stats[c]->tx_count[CLASS]);
        // algorithmic bytes: int16 coefficients read + pixels read-modify-written
        bytes += 2 * s.stats[c]->tx_coefs[CLASS] + 2 * s.stats[c]->tx_pixels[CLASS];
    }
    if (!max_count) return;
    constexpr int BPW = 64 / TxLds<W, H>::THREADS;
    ProfScope ps(prof, name, bytes, st);
    hipLaunchKernelGGL((itx_batch_kernel<W, H, CLASS>), dim3((max_count + BPW - 1) / BPW, 1, s.batch.n), dim3(64), 0, st, s.batch);
}

void launch_itx(const SubmitInfo &s, hipStream_t st, Prof *prof)
{
    // one launch per transform size that occurs in the batch (names = what rocprofv3 prints)
#define EA1_TXC(W_, H_, C_) launch_class<W_, H_>(s, st, prof, "itx_batch_kernel<" #W_ "," #H_ "," #C_ ">")
    EA1_TXC(4, 4, 0); EA1_TXC(8, 8, 1); EA1_TXC(16, 16, 2); EA1_TXC(32, 32, 3); EA1_TXC(64, 64, 4);
    EA1_TXC(4, 8, 5); EA1_TXC(8, 4, 6); EA1_TXC(8, 16, 7); EA1_TXC(16, 8, 8); EA1_TXC(16, 32, 9); EA1_TXC(32, 16, 10);
    EA1_TXC(32, 64, 11); EA1_TXC(64, 32, 12); EA1_TXC(4, 16, 13); EA1_TXC(16, 4, 14); EA1_TXC(8, 32, 15); EA1_TXC(32, 8, 16);
    EA1_TXC(16, 64, 17); EA1_TXC(64, 16, 18);
#undef EA1_TXC
}

// ---- DSP-level entry: any of the 19 AV1 transform sizes on a bare plane (parity tests) ----
template <int W, int H>
__global__ void __launch_bounds__(64)
itx_items_kernel(const TxItem *__restrict__ items, int count, const int16_t *__restrict__ coefs, uint8_t *plane, int stride)
{
    using T = TxLds<W, H>;
    constexpr int BPW = 64 / T::THREADS;
    __shared__ __attribute__((aligned(16))) int16_t lds[BPW * T::TOTAL];
    const int g = threadIdx.x / T::THREADS, sub = threadIdx.x % T::THREADS;
    const int idx = blockIdx.x * BPW + g;
    const bool active = idx < count;
    const TxItem it = items[active ? idx : 0];
    uint8_t *dst = plane + size_t(it.y) * stride + it.x;
    itx::tx_block_add<W, H, 32>(lds + g * T::TOTAL, coefs + it.coef_off, dst, stride, it.txtp, sub, active);
}

template <int W, int H>
static void launch_items(const TxItem *items, int count, const int16_t *coefs, uint8_t *plane, int stride, hipStream_t st)
{
    constexpr int BPW = 64 / TxLds<W, H>::THREADS;
    hipLaunchKernelGGL((itx_items_kernel<W, H>), dim3((count + BPW - 1) / BPW), dim3(64), 0, st, items, count, coefs, plane, stride);
}

int launch_dsp_itx_add(uint8_t *plane, size_t stride, int w, int h, const int16_t *d_coefs, const uint32_t *pos,
                       const uint8_t *txtp, int count, hipStream_t st)
{
    const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
    TxItem *h_items = static_cast<TxItem *>(malloc(sizeof(TxItem) * size_t(count)));
    if (!h_items) return fail(EA1_ENOMEM, "out of memory");
    for (int i = 0; i < count; i++)
        h_items[i] = TxItem{uint32_t(i) * uint32_t(sw * sh), uint16_t(pos[2 * i]), uint16_t(pos[2 * i + 1]), 0, txtp[i], 0, 0, 0};
    TxItem *d_items = nullptr;
    hipError_t e = hipMalloc(&d_items, sizeof(TxItem) * size_t(count));
    if (e == hipSuccess) e = hipMemcpyAsync(d_items, h_items, sizeof(TxItem) * size_t(count), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        free(h_items);
        if (d_items) (void)hipFree(d_items);
        return fail(EA1_EHIP, "dsp_itx_add: %s", hipGetErrorString(e));
    }
    int rc = EA1_OK;
#define EA1_CASE(W_, H_) else if (w == W_ && h == H_) launch_items<W_, H_>(d_items, count, d_coefs, plane, int(stride), st)
    if (false) {}
    EA1_CASE(4, 4); EA1_CASE(8, 8); EA1_CASE(16, 16); EA1_CASE(32, 32); EA1_CASE(64, 64);
    EA1_CASE(4, 8); EA1_CASE(8, 4); EA1_CASE(8, 16); EA1_CASE(16, 8); EA1_CASE(16, 32); EA1_CASE(32, 16);
    EA1_CASE(32, 64); EA1_CASE(64, 32); EA1_CASE(4, 16); EA1_CASE(16, 4); EA1_CASE(8, 32); EA1_CASE(32, 8);
    EA1_CASE(16, 64); EA1_CASE(64, 16);
    else rc = fail(EA1_EINVAL, "dsp_itx_add: %dx%d is not an AV1 transform size", w, h);
#undef EA1_CASE
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    free(h_items);
    (void)hipFree(d_items);
    if (rc == EA1_OK && e != hipSuccess) rc = fail(EA1_EHIP, "dsp_itx_add kernel: %s", hipGetErrorString(e));
    return rc;
}

}  // namespace ea1
