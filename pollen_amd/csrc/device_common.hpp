// Shared between the C-ABI translation unit and the HIP translation unit.
#pragma once
#include <cstdlib>
#include <string>

// Environment variables reach this library in two kinds.  The handful a user may set -- FLATGFA_DEPTH_PATH, FLATGFA_MALL_MB,
// FLATGFA_PACKED, FLATGFA_BUCKET_GB, FLATGFA_UPLOAD_THREADS, FLATGFA_PARSE_THREADS, FLATGFA_TIMING, FLATGFA_CHECK_NO_CLAIM,
// FLATGFA_SHARD_FORCE_RCCL, FLATGFA_NO_WARM, FLATGFA_KEEP_HOST_MEMORY (the CLI and the Python package): INTEGRATION.md section 6 -- are read with getenv where they apply; none of them
// changes a result.  Everything else is a TEST HOOK: the parity suite forces every device path of the shipped library
// (tests/test_gpu_depth.py: `device_path`, tools/fuzz_gpu.py: ENVS) by shaping the plan -- piece sizes, window sizes, bucket
// capacities, which kernel walks which path -- and reads two diagnostics (FLATGFA_SCAN_TIME, FLATGFA_ACC_TIME).  They go through
// this one function so that they can be told apart (and found: `grep test_hook`); they change which kernels run, never
// what they count.
#ifndef FGFA_TEST_HOOK_DEFINED
#define FGFA_TEST_HOOK_DEFINED
inline const char *test_hook(const char *name) { return std::getenv(name); }
#endif

namespace fgfa_dev {
void set_error(const std::string &s);
const char *last_error();
}  // namespace fgfa_dev
// An empty launch on `stream`: what makes the runtime load this library's code object (flatgfa_warm_device).
namespace fgfa_dev {
void warm_launch(void *stream);
}
