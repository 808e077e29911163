// TEST HARNESS (tests/ only): a minimal stand-in for <hip/hip_runtime.h> -- just the host API the C-ABI translation units
// (sba_shim / sba_transport / sba_stages / sba_batch .cpp) use -- so that their HANDLE LOGIC (error paths, poisoning,
// destroy) can be compiled with plain g++ and driven on the CPU against a mock device (tests/harness/wedge_harness.cpp:
// "device memory" is host memory, kernels are stubs, a stream can be wedged so that it never drains).
// Not a HIP implementation and never part of the product: libsba_hip.so is always built against the real runtime.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorNoDevice = 100, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorUnknown = 999 };
typedef struct fake_hip_stream* hipStream_t;
typedef struct fake_hip_event* hipEvent_t;
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; };
struct hipIpcMemHandle_t { char reserved[64]; };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
enum : unsigned { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000u,
                  hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3, hipIpcMemLazyEnablePeerAccess = 1 };

const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int*);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceProperties(hipDeviceProp_t*, int);
hipError_t hipDeviceGetAttribute(int*, hipDeviceAttribute_t, int);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void**, size_t);
hipError_t hipExtMallocWithFlags(void**, size_t, unsigned);
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostGetDevicePointer(void**, void*, unsigned);
hipError_t hipHostFree(void*);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
constexpr unsigned hipEventDisableTiming = 2;
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*);
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned);
hipError_t hipIpcCloseMemHandle(void*);
hipError_t hipDeviceEnablePeerAccess(int, unsigned);
