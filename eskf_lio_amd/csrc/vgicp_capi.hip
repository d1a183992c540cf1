// vgicp_capi.hip — the C ABI declared in include/vgicp_hip.h (host side of the HIP module).
//
// Owns: the device context, the voxel-table mirror of LocalMap (growth / rehash policy), the resident
// scan, the iteration launch schedule of ICP::align (reference src/Registration.cpp:15-28) and the
// optional RCCL communicator that merges the normal equations across GPUs (the cross-device form of
// the thread merge at reference src/Registration.cpp:71-75).
// No PyTorch, no oracle, no CPU fallback: without a gfx950 device every compute entry point fails
// with VGICP_ERR_NO_DEVICE / VGICP_ERR_HIP.
//
// ONE translation unit, cut by concern into the files included below (round 6; it was 3 200 lines in one file): the
// helpers share an unnamed namespace, so the parts are textual pieces of this file, in this order — each needs what the
// ones before it define — not separate objects.
#include <immintrin.h>

#include <array>
#include <unordered_map>
#include "vgicp_context.h"

namespace vgicp {
thread_local uint64_t g_copy_ops = 0, g_sync_ops = 0;
thread_local std::string g_create_error;
thread_local std::string g_stage_error;
thread_local uint64_t g_stage_error_ctx = 0;
std::atomic<uint64_t> g_context_ids{0};
}  // namespace vgicp
#include "vgicp_capi_memory.inl"
#include "vgicp_capi_align.inl"
#include "vgicp_capi_context.inl"
#include "vgicp_capi_map.inl"
#include "vgicp_capi_upload.inl"
#include "vgicp_capi_registration.inl"
#include "vgicp_capi_prepare.inl"
#include "vgicp_capi_peers.inl"
#include "vgicp_capi_multi_support.inl"
