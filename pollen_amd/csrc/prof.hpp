// Per-kernel HIP-event timing (flatgfa_dev_profile_enable / _read in include/flatgfa.h).
#pragma once
#include <hip/hip_runtime_api.h>

namespace fgfa_dev {

struct ProfRec {
    const char *name;
    hipEvent_t a, b;
    int device;  // (the events' device: they go back to its pool)
};
bool prof_enabled();
void prof_push(const ProfRec &r);
// Events are taken from (and, by flatgfa_dev_profile_read, returned to) a pool: creating a pair per
// kernel cost more than recording it, and it is the timed region that pays.
hipEvent_t prof_event_get();
void prof_event_put(hipEvent_t e);

// Brackets the launches issued during its lifetime with two events on their stream.
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfRec r;
    ProfScope(const char *name, hipStream_t stream) : s(stream), on(prof_enabled()) {
        if (on) {
            r.name = name;
            r.device = 0;
            (void)hipGetDevice(&r.device);
            r.a = prof_event_get();
            r.b = prof_event_get();
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
