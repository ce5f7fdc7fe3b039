// bifrost-named adapters: same names and argument shapes as the reference's ctypes call
// sites (`from bifrost.libbifrost import _bf`), taking BFarray-like structs.  Only the data
// pointers (and, where needed, element counts) are read from the structs; sizes come from
// the configured contexts (xGPU's were compile-time constants, install_xgpu.sh:5).
#include "xeng_common.h"

using namespace xeng;

static inline bool bad(const XENGarray* a) { return a == nullptr || a->data == nullptr; }

extern "C" {

int bfXgpuInitialize(XENGarray* in, XENGarray* out, int gpu_dev) {
    (void)in; (void)out;  // dummies at the reference call site too (corr_block.py:249-252)
    return xengXgpuInitialize(gpu_dev);
}
int bfXgpuKernel(XENGarray* in, XENGarray* out, int doDump) {
    if (bad(in) || bad(out)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuKernel: null array");
    return xengXgpuKernel(in->data, out->data, doDump);
}
int bfXgpuCorrelate(XENGarray* in, XENGarray* out, int doDump) {
    if (bad(in) || bad(out)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuCorrelate: null array");
    return xengXgpuCorrelate(in->data, out->data, doDump);
}
int bfXgpuGetOrder(XENGarray* antpol_to_input, XENGarray* antpol_to_bl, XENGarray* is_conj) {
    if (bad(antpol_to_input) || bad(antpol_to_bl) || bad(is_conj)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuGetOrder: null array");
    return xengXgpuGetOrder((const int32_t*)antpol_to_input->data, (int32_t*)antpol_to_bl->data, (int32_t*)is_conj->data);
}
int bfXgpuSubSelect(XENGarray* in, XENGarray* out, XENGarray* vismap, XENGarray* conj, int nchan_sum, int unused) {
    (void)unused;
    if (bad(in) || bad(out) || bad(vismap) || bad(conj)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuSubSelect: null array");
    if (vismap->ndim < 1) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuSubSelect: vismap has no shape");
    return xengXgpuSubSelect(in->data, out->data, (const int32_t*)vismap->data, (const int32_t*)conj->data,
                             (int)vismap->shape[0], nchan_sum);
}
int bfXgpuReorder(XENGarray* in, XENGarray* out, XENGarray* baselines, XENGarray* is_conj) {
    if (bad(in) || bad(out) || bad(baselines) || bad(is_conj)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfXgpuReorder: null array");
    return xengXgpuReorder(in->data, out->data, (const int32_t*)baselines->data, (const int32_t*)is_conj->data);
}
int bfBeamformInitialize(int gpu, int ninput, int nchan, int ntime, int nbeam, int ntime_blocks) {
    return xengBeamformInitialize(gpu, ninput, nchan, ntime, nbeam, ntime_blocks);
}
int bfBeamformRun(XENGarray* in, XENGarray* out, XENGarray* weights) {
    if (bad(in) || bad(out) || bad(weights)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfBeamformRun: null array");
    return xengBeamformRun(in->data, out->data, weights->data);
}
int bfBeamformIntegrate(XENGarray* in, XENGarray* out, int ntime_sum) {
    if (bad(in) || bad(out)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfBeamformIntegrate: null array");
    return xengBeamformIntegrate(in->data, out->data, ntime_sum);
}
int bfBeamformIntegrateSingleBeam(XENGarray* in, XENGarray* out, int ntime_sum, int beam_id) {
    if (bad(in) || bad(out)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bfBeamformIntegrateSingleBeam: null array");
    return xengBeamformIntegrateSingleBeam(in->data, out->data, ntime_sum, beam_id);
}

}  // extern "C"
