// The reference's native module on the hot path, in the reference's own FORM: a torch C++ extension built with
// torch.utils.cpp_extension (model/train.py:122-125 JIT-builds src/utils_c/pack2bit.cu the same way), exporting the same two
// functions with the same Tensor signatures (pack2bit.cu:144-147) -- over libnadm.so's C ABI (include/nadm.h) instead of CUDA code
// of its own.  Contract as there: the CALLER allocates both tensors (train.py:121, neural_admixture.py:377,405), shape / device
// mismatches raise through TORCH_CHECK with the reference's messages (pack2bit.cu:66-76,121-130), the launches go to the legacy
// default stream and both calls block until the result is complete (pack2bit.cu:115,141).  Difference, invisible to the caller: the
// matrix is packed on the host, so 2 bits per genotype cross PCIe instead of 8 (pack2bit.cu:79-115 ships unpacked bytes in
// 1024-row chunks and packs on the device).
//
// Built by __graft_entry__.build() into neural-admixture_amd/csrc/ext/_pack2bit.so; neural_admixture_amd.pack2bit imports it.
#include <torch/extension.h>
#include <torch/cuda.h>
#include <c10/core/DeviceGuard.h>
#include <algorithm>

extern "C" {
int nadm_pack2bit_host(const uint8_t* g_host, uint8_t* out_host, int64_t N, int64_t M, int64_t ld);
int nadm_unpack2bit(const uint8_t* in_dev, uint8_t* out_dev, int64_t rows, int64_t M, int64_t ld, void* stream);
const char* nadm_last_error(void);
}

namespace {

constexpr int64_t kStageBytes = int64_t(64) << 20;          // pinned staging buffer, sized by bytes and kept for the process

void pack2bit_cpu_to_gpu(torch::Tensor input_cpu, torch::Tensor output_gpu) {
    TORCH_CHECK(input_cpu.device().is_cpu(), "Input tensor must be on CPU");
    TORCH_CHECK(output_gpu.device().is_cuda(), "Output tensor must be on CUDA device");
    TORCH_CHECK(input_cpu.dim() == 2 && output_gpu.dim() == 2, "pack2bit_cpu_to_gpu expects 2-D tensors");
    TORCH_CHECK(input_cpu.scalar_type() == torch::kUInt8 && output_gpu.scalar_type() == torch::kUInt8, "pack2bit_cpu_to_gpu expects uint8 tensors");
    const int64_t N = input_cpu.size(0), M = input_cpu.size(1), packed_cols = (M + 3) / 4;
    TORCH_CHECK(output_gpu.size(0) == N, "Output tensor row dimension mismatch");
    TORCH_CHECK(output_gpu.size(1) == packed_cols, "Output tensor column dimension mismatch");
    TORCH_CHECK(output_gpu.is_contiguous(), "Output tensor must be contiguous");
    if (N == 0 || M == 0) return;
    const torch::Tensor src = input_cpu.contiguous();
    const int64_t rows = std::max<int64_t>(1, std::min<int64_t>(N, kStageBytes / packed_cols));
    // kept for the process and never destroyed: a static Tensor's destructor would free pinned memory after the HIP runtime is gone
    static torch::Tensor* stage = new torch::Tensor();
    if (!stage->defined() || stage->numel() < rows * packed_cols)
        *stage = torch::empty({rows * packed_cols}, torch::dtype(torch::kUInt8).pinned_memory(true));
    torch::Tensor st = stage->narrow(0, 0, rows * packed_cols).view({rows, packed_cols});
    for (int64_t s = 0; s < N; s += rows) {
        const int64_t e = std::min(N, s + rows);
        TORCH_CHECK(nadm_pack2bit_host(src.data_ptr<uint8_t>() + s * M, st.data_ptr<uint8_t>(), e - s, M, packed_cols) == 0,
                    "pack2bit_cpu_to_gpu: ", nadm_last_error());
        output_gpu.narrow(0, s, e - s).copy_(st.narrow(0, 0, e - s), /*non_blocking=*/false);      // the staging buffer is reused
    }
    torch::cuda::synchronize(output_gpu.device().index());
}

void unpack2bit_gpu_to_gpu(torch::Tensor input_gpu, torch::Tensor output_gpu) {
    TORCH_CHECK(input_gpu.device().is_cuda(), "Input tensor must be on CUDA device");
    TORCH_CHECK(output_gpu.device().is_cuda(), "Output tensor must be on CUDA device");
    TORCH_CHECK(input_gpu.device() == output_gpu.device(), "Input and Output tensors must be on the same CUDA device");
    TORCH_CHECK(input_gpu.dim() == 2 && output_gpu.dim() == 2, "unpack2bit_gpu_to_gpu expects 2-D tensors");
    TORCH_CHECK(input_gpu.scalar_type() == torch::kUInt8 && output_gpu.scalar_type() == torch::kUInt8, "unpack2bit_gpu_to_gpu expects uint8 tensors");
    const int64_t N = output_gpu.size(0), M = output_gpu.size(1), packed_cols = (M + 3) / 4;
    TORCH_CHECK(input_gpu.size(0) == N, "Input tensor row dimension mismatch");
    TORCH_CHECK(input_gpu.size(1) == packed_cols, "Input tensor column dimension mismatch based on output shape");
    TORCH_CHECK(output_gpu.is_contiguous(), "Output tensor must be contiguous");
    if (N == 0 || M == 0) return;
    const c10::DeviceGuard on_device(output_gpu.device());      // the launch goes to the tensors' device, whichever one is current
    const torch::Tensor src = input_gpu.contiguous();           // a DataLoader batch of gathered rows is contiguous already
    torch::cuda::synchronize(output_gpu.device().index());      // whatever produced the batch on torch's streams is complete ...
    TORCH_CHECK(nadm_unpack2bit(src.data_ptr<uint8_t>(), output_gpu.data_ptr<uint8_t>(), N, M, packed_cols, /*stream=*/nullptr) == 0,
                "unpack2bit_gpu_to_gpu: ", nadm_last_error());  // ... the launch goes to the legacy default stream (pack2bit.cu:139) ...
    torch::cuda::synchronize(output_gpu.device().index());      // ... and the call blocks (pack2bit.cu:141)
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("pack2bit_cpu_to_gpu", &pack2bit_cpu_to_gpu, "Pack 2-bit data from CPU uint8 tensor to GPU uint8 tensor");
    m.def("unpack2bit_gpu_to_gpu", &unpack2bit_gpu_to_gpu, "Unpack 2-bit data from GPU uint8 tensor to GPU uint8 tensor");
}
