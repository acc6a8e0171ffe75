// ptx_context_access.h -- narrow accessors into ptx_context for translation units other than ptx_api.cpp.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string>

struct ptx_context;

namespace ptx {
uint64_t ctx_pixels(ptx_context *c);
int ctx_device(ptx_context *c);
void *ctx_albedo(ptx_context *c);
hipStream_t ctx_stream(ptx_context *c);
int ctx_fail(ptx_context *c, int code, const std::string &msg);
// ptx_rccl.cpp: contexts alive per device; the cached RCCL communicators of a device set go when the last
// context on one of its devices is destroyed
void rccl_context_created(int device);
void rccl_context_destroyed(int device);
}  // namespace ptx
