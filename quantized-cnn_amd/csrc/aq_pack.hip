// Bit-packing of the hard-mode codes: idx = argmax s_c (adp_qtz_fan_test.py:304) takes values
// < L <= 128, i.e. `bits` = ceil(log2 L) bits (5 for the 31/32-level layers), so a layer's weights
// are stored as a little-endian bit stream of codes plus its L-entry codebook q_level.
// Code i occupies stream bits [i*bits, (i+1)*bits), least significant bit first.  Eight codes make
// exactly `bits` bytes, so one thread handles one such group without touching its neighbours' bytes.
#include "aq_common.h"
#include "../../include/aq_hip.h"

namespace aq {

__global__ __launch_bounds__(256) void pack_codes_kernel(const uint8_t *__restrict__ idx, long long n, int bits, uint8_t *__restrict__ out,
                                                         long long out_bytes)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i0 = g * 8;
    if (i0 >= n) return;
    unsigned long long word = 0;
    const unsigned mask = (1u << bits) - 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned c = i0 + k < n ? idx[i0 + k] : 0u;
        word |= (unsigned long long)(c & mask) << (k * bits);
    }
    uint8_t *o = out + g * bits;
    for (int b = 0; b < bits; ++b)
        if (g * bits + b < out_bytes) o[b] = (uint8_t)(word >> (8 * b));
}

__global__ __launch_bounds__(256) void unpack_codes_kernel(const uint8_t *__restrict__ packed, long long n, int bits, uint8_t *__restrict__ idx,
                                                           long long in_bytes)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i0 = g * 8;
    if (i0 >= n) return;
    unsigned long long word = 0;
    const uint8_t *p = packed + g * bits;
    for (int b = 0; b < bits; ++b)
        if (g * bits + b < in_bytes) word |= (unsigned long long)p[b] << (8 * b);
    const unsigned mask = (1u << bits) - 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (i0 + k < n) idx[i0 + k] = (uint8_t)((word >> (k * bits)) & mask);
}

} // namespace aq

extern "C" {

size_t aq_packed_bytes(long long n, int bits)
{
    if (n < 0 || bits < 1 || bits > 8) return 0;
    return (size_t)((n * bits + 7) / 8);
}

int aq_pack_codes(const uint8_t *idx, long long n, int bits, uint8_t *out, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && bits >= 1 && bits <= 8, -EINVAL, "aq_pack_codes: n=%lld bits=%d (bits must be 1..8)", n, bits);
    if (n == 0) return 0;
    AQ_REQUIRE(idx && out, -EFAULT, "aq_pack_codes: NULL pointer");
    const long long groups = (n + 7) / 8;
    AQ_REQUIRE((groups + 255) / 256 < (1LL << 31), -E2BIG, "aq_pack_codes: too many codes");
    hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), idx, n, bits, out,
                       (long long)aq_packed_bytes(n, bits));
    return check_launch("aq_pack_codes");
}

int aq_unpack_codes(const uint8_t *packed, long long n, int bits, uint8_t *idx, void *stream)
{
    using namespace aq;
    AQ_REQUIRE(n >= 0 && bits >= 1 && bits <= 8, -EINVAL, "aq_unpack_codes: n=%lld bits=%d (bits must be 1..8)", n, bits);
    if (n == 0) return 0;
    AQ_REQUIRE(idx && packed, -EFAULT, "aq_unpack_codes: NULL pointer");
    const long long groups = (n + 7) / 8;
    AQ_REQUIRE((groups + 255) / 256 < (1LL << 31), -E2BIG, "aq_unpack_codes: too many codes");
    hipLaunchKernelGGL(unpack_codes_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), packed, n, bits,
                       idx, (long long)aq_packed_bytes(n, bits));
    return check_launch("aq_unpack_codes");
}

} // extern "C"
