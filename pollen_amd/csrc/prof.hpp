// Per-kernel HIP-event timing (flatgfa_dev_profile_enable / _read in include/flatgfa.h).
#pragma once
#include <hip/hip_runtime_api.h>

namespace fgfa_dev {

struct ProfRec {
    const char *name;
    hipEvent_t a, b;
};
bool prof_enabled();
void prof_push(const ProfRec &r);

// Brackets the launches issued during its lifetime with two events on their stream.
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfRec r;
    ProfScope(const char *name, hipStream_t stream) : s(stream), on(prof_enabled()) {
        if (on) {
            r.name = name;
            (void)hipEventCreate(&r.a);
            (void)hipEventCreate(&r.b);
            (void)hipEventRecord(r.a, s);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(r.b, s);
            prof_push(r);
        }
    }
};

}  // namespace fgfa_dev
