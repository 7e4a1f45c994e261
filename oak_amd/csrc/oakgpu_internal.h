// oak_amd/csrc/oakgpu_internal.h -- shared between the translation units of liboakgpu.so.
#pragma once
struct oakgpu_ctx;
int oakgpu_fail_hip(int hip_error, const char *what); // records hipGetErrorString, returns the code
int oakgpu_fail_msg(const char *what);                 // records the message, returns -1
int oakgpu_ctx_device(const oakgpu_ctx *ctx);
void *oakgpu_ctx_stream(const oakgpu_ctx *ctx);        // hipStream_t
