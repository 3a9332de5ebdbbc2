// A second, thinner binding of ONE C-ABI entry: sympa_model_forward (include/sympa_hip.h), for the per-batch loops of the
// reference (Runner.train_epoch / evaluate call the model once per batch, sympa/runner.py:98-101,126-131).  Through ctypes
// one Model.forward costs ~15 us of host time against a 7 us kernel (18 arguments converted one by one, the output
// allocated and the stream looked up from Python); here the tensors arrive as at::Tensor, the checks, the output allocation
// and the current HIP stream are C++ (SURVEY 7 step 3 named this binding as the alternative to ctypes).
//
// This file is PLUMBING: it contains no arithmetic and no kernel; it calls the C-ABI through a function pointer that
// sympa_amd/_lib.py takes from the SAME libsympa_hip.so the ctypes binding loaded (so SYMPA_HIP_LIB variant builds work),
// typed by the header's own prototype -- a signature drift between header and binding is a compile error here.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include "../../include/sympa_hip.h"

namespace {

using model_forward_fn = decltype(&sympa_model_forward);
using last_error_fn = decltype(&sympa_last_error);
model_forward_fn g_model_forward = nullptr;
last_error_fn g_last_error = nullptr;

void bind(uint64_t model_forward_addr, uint64_t last_error_addr) {
    g_model_forward = reinterpret_cast<model_forward_fn>(static_cast<uintptr_t>(model_forward_addr));
    g_last_error = reinterpret_cast<last_error_fn>(static_cast<uintptr_t>(last_error_addr));
}

// Model.forward without autograd (model.py:16-41): table [N,2,n,n] fp64, triplets [b, >=2] int64 with unit column stride,
// weights [n] fp64 or None, scale [1] fp64 or None, status int32[2]; returns (or fills `out`) [b] fp64.
at::Tensor model_forward(const at::Tensor& table, const at::Tensor& triplets, int64_t model, int64_t metric,
                         const c10::optional<at::Tensor>& weights, const c10::optional<at::Tensor>& scale, double scale_coef,
                         double eps, const at::Tensor& status, int64_t flags, const c10::optional<at::Tensor>& out_opt) {
    TORCH_CHECK(g_model_forward != nullptr, "sympa_amd._fast is not bound to libsympa_hip.so");
    TORCH_CHECK(table.is_cuda() && triplets.is_cuda(),
                "the Siegel-distance path runs on the GPU only (HIP kernels, no CPU fallback)");
    TORCH_CHECK_TYPE(table.scalar_type() == at::kDouble, "table must be float64");
    TORCH_CHECK_TYPE(triplets.scalar_type() == at::kLong && triplets.dim() == 2 && triplets.size(1) >= 2,
                     "triplets must be an int64 [b, >=2] tensor (src, dst[, graph_distance])");
    TORCH_CHECK_VALUE(table.dim() == 4 && table.size(1) == 2 && table.size(2) == table.size(3) && table.is_contiguous(),
                      "table must be a contiguous [N,2,n,n] tensor");
    TORCH_CHECK_VALUE(triplets.stride(1) == 1, "triplets must have unit column stride");
    TORCH_CHECK_VALUE(triplets.get_device() == table.get_device(), "table and triplets live on different devices");
    const int64_t b = triplets.size(0);
    const int n = static_cast<int>(table.size(2));
    at::Tensor out = out_opt.has_value() ? *out_opt : at::empty({b}, table.options());
    if (out_opt.has_value())
        TORCH_CHECK_VALUE(out.is_cuda() && out.scalar_type() == at::kDouble && out.is_contiguous() && out.numel() >= b,
                          "out must be a contiguous float64 device tensor of at least b elements");
    if (b == 0) return out;
    const double* w = nullptr;
    if (weights.has_value()) {
        TORCH_CHECK_VALUE(weights->is_cuda() && weights->scalar_type() == at::kDouble && weights->is_contiguous() &&
                              weights->numel() == n, "wsum weights: contiguous float64 [n] on the device");
        w = weights->data_ptr<double>();
    }
    const double* sc = nullptr;
    if (scale.has_value()) {
        TORCH_CHECK_VALUE(scale->is_cuda() && scale->scalar_type() == at::kDouble && scale->numel() >= 1,
                          "scale: float64 on the device");
        sc = scale->data_ptr<double>();
    }
    const int64_t stride = b > 1 ? triplets.stride(0) : triplets.size(1);
    const int64_t* tp = triplets.data_ptr<int64_t>();
    const c10::hip::HIPStream stream = c10::hip::getCurrentHIPStream(table.get_device());
    const c10::DeviceGuard guard(table.device());        // the launch goes to the table's device
    const int rc = g_model_forward(table.data_ptr<double>(), table.size(0), n, tp, stride, tp + 1, stride, b,
                                   static_cast<int>(model), static_cast<int>(metric), w, eps, sc, scale_coef,
                                   out.data_ptr<double>(), status.data_ptr<int32_t>(), static_cast<int>(flags), stream.stream());
    if (rc != 0) {
        const char* msg = g_last_error != nullptr ? g_last_error() : "";
        if (rc == SYMPA_ERR_UNSUPPORTED_DIMS) TORCH_CHECK(false, "sympa_hip: unsupported dims: ", msg);
        TORCH_CHECK_VALUE(false, "sympa_hip error ", rc, ": ", msg);
    }
    return out;
}

}  // namespace

PYBIND11_MODULE(_fast, m) {
    m.doc() = "thin torch binding of C-ABI sympa_model_forward (include/sympa_hip.h)";
    m.def("bind", &bind, "addresses of sympa_model_forward and sympa_last_error in the loaded libsympa_hip.so");
    m.def("model_forward", &model_forward);
}
