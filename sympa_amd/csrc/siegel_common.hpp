// Shared host/device declarations of the translation units of libsympa_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/sympa_hip.h"
#include "siegel_math.hpp"
#include "siegel_gather.hpp"

namespace sympa_hip {

constexpr int BLOCK = 256;

struct DistArgs {
    const double* base1;   // z1 [b,2,n,n]   or the table
    const double* base2;   // z2 [b,2,n,n]   or the table
    const int64_t* idx1;   // nullptr -> row i
    const int64_t* idx2;
    int64_t idx1_stride;
    int64_t idx2_stride;
    int64_t num_rows;      // rows addressable through idx (bounds check)
    int64_t b;
    const double* metric_w;
    const double* scale;   // device scalar or nullptr
    double inv_scale_coef;
    double inv_eps;        // 1 / eps, computed on the host
    double* out;
    double* vvd;
    int32_t* status;
    int metric;
    int flags;
    int64_t ap_cols;       // > 0: all-pairs mode, pair p = (ap_row0 + p / ap_cols, p % ap_cols), idx1/idx2 unused
    int64_t ap_row0;
};

// thread-local message behind sympa_last_error()
char* last_error_buffer();
int fail(int code, const char* msg);
int validate(const DistArgs& a, int model, int n);

// siegel_coop.hip: 9 <= n <= 16, sixteen lanes per pair
int launch_siegel_coop(const DistArgs& a, int n, int model, hipStream_t s);

}  // namespace sympa_hip
