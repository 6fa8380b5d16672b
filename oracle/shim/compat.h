/* Force-included ahead of every reference translation unit (oracle/Makefile, -include).
 * - glibc's <unistd.h> defines R_OK for access(); it would clobber pk::R_OK (result.h:7).
 * - sprintf_s is MSVC-only (utils.h:86-91, SET_THREAD_NAME). */
#pragma once
#include <unistd.h>
#undef R_OK
#include <cstddef>
#include <cstdio>
template <size_t N> inline int sprintf_s(char (&dst)[N], const char* src) { return snprintf(dst, N, "%s", src); }
