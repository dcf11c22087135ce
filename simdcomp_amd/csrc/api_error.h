// The per-thread error text behind simdcomp_hip_last_error(), shared by the translation units of the library.
#pragma once

namespace sc {
// Records `what` as the calling thread's last error and returns `code`.
int api_fail(int code, const char *what);
}  // namespace sc
