// Shared between the C-ABI translation unit and the HIP translation unit.
#pragma once
#include <string>

namespace fgfa_dev {
void set_error(const std::string &s);
const char *last_error();
}  // namespace fgfa_dev
// An empty launch on `stream`: what makes the runtime load this library's code object (flatgfa_warm_device).
namespace fgfa_dev {
void warm_launch(void *stream);
}
