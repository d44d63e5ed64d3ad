// Shared between the C-ABI translation unit and the HIP translation unit.
#pragma once
#include <string>

namespace fgfa_dev {
void set_error(const std::string &s);
const char *last_error();
}  // namespace fgfa_dev
