// Host-only exercise of libemavfi's C-ABI for the sanitizer build: every entry that does
// its work - or refuses its arguments - on the host: the *_bytes queries, emavfi_supported, emavfi_forward_launches, the argument
// guards of the forward / stage entries, the blob header check on host memory, the switch word.  No kernel is launched and no GPU
// is needed; the library is built with -fsanitize=address,undefined (csrc/Makefile, target `asan`) and this program with it, so
// plan building, workspace carving, string handling and the guards run under ASan + UBSan on the CPU box.
// Test infrastructure: tests/test_cabi_cpu.py::test_host_side_runs_clean_under_asan_ubsan builds and runs it.
#include "../../include/emavfi.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_fail = 0;
#define CHECK(cond)                                                                             \
    do {                                                                                        \
        if (!(cond)) { fprintf(stderr, "host_check: %s:%d: %s  [last error: %s]\n", __FILE__, __LINE__, #cond, emavfi_last_error()); ++g_fail; } \
    } while (0)

static std::vector<unsigned char> host_blob(int mid, int dtype, uint32_t tag, uint32_t version)
{
    const size_t total = emavfi_packed_bytes(3, mid, 3, dtype);
    std::vector<unsigned char> b(total, 0);
    uint32_t seed = 12345u;
    for (size_t i = 256; i < total; ++i) { seed = seed * 1664525u + 1013904223u; b[i] = (unsigned char)(seed >> 24); }
    uint64_t sum = 0;
    for (size_t i = 0; i < (total - 256) / 4; ++i) {
        uint32_t w;
        memcpy(&w, &b[256 + 4 * i], 4);
        sum += ((uint64_t)w + 0x9E3779B9ull) * (2 * (uint64_t)i + 1);
    }
    uint32_t h[16] = {0};
    memcpy(h, "EMAVFIPK", 8);
    h[2] = version; h[3] = 256; h[4] = 3; h[5] = (uint32_t)mid; h[6] = 3; h[7] = (uint32_t)dtype; h[8] = tag;
    memcpy(&b[0], h, 64);
    const uint64_t tb = total;
    memcpy(&b[40], &tb, 8);
    memcpy(&b[48], &sum, 8);
    return b;
}

int main()
{
    CHECK(emavfi_version() == EMAVFI_VERSION);
    CHECK(emavfi_param_count(3) == 40 && emavfi_param_count(1) == 28);
    const int dtypes[4] = {EMAVFI_F32, EMAVFI_BF16, EMAVFI_F16, EMAVFI_AMP16};
    const int modes[5] = {EMAVFI_F32, EMAVFI_BF16, EMAVFI_F16, EMAVFI_AMP16, EMAVFI_F32X3};   // (the split mode has a forward and a conv3x3 stage entry only)
    for (int mid : {8, 16, 32, 64})
        for (int dt : modes)
            for (int nb : {1, 2, 3, 8}) {
                CHECK(emavfi_supported(3, mid, nb, dt) == EMAVFI_OK);
                CHECK(emavfi_packed_bytes(3, mid, nb, dt) > 256);
                for (int B : {1, 3})
                    for (int H : {1, 23, 720})
                        for (int W : {1, 37, 1280}) CHECK(emavfi_workspace_bytes(3, mid, nb, B, H, W, dt) > 0);
            }
    CHECK(emavfi_supported(3, 7, 3, EMAVFI_F32) == EMAVFI_E_UNSUPPORTED && strstr(emavfi_last_error(), "multiple of 8"));
    CHECK(emavfi_supported(4, 64, 3, EMAVFI_F32) == EMAVFI_E_UNSUPPORTED);
    CHECK(emavfi_supported(3, 64, 9, EMAVFI_F32) == EMAVFI_E_UNSUPPORTED && emavfi_supported(3, 64, 0, EMAVFI_BF16) == EMAVFI_E_UNSUPPORTED);
    CHECK(emavfi_supported(3, 64, 3, 7) == EMAVFI_E_UNSUPPORTED && emavfi_supported(3, 24, 3, EMAVFI_BF16) == EMAVFI_E_UNSUPPORTED);
    CHECK(emavfi_packed_bytes(3, 64, 0, EMAVFI_F32) == 0 && emavfi_workspace_bytes(3, 64, 3, 0, 8, 8, EMAVFI_F32) == 0);
    CHECK(emavfi_workspace_bytes(3, 64, 3, 1, -5, 8, EMAVFI_F32) == 0);

    // launch enumeration: exact buffers, a names buffer that is too small, capacity too small, null outputs
    for (int dt : modes) {
        const int n = emavfi_forward_launches(3, 64, 3, 2, 96, 128, dt, nullptr, 0, nullptr, nullptr, 0);
        CHECK(n >= 12 && n <= 40);
        std::vector<char> names(128 * (size_t)n);
        std::vector<double> fl((size_t)n), by((size_t)n);
        CHECK(emavfi_forward_launches(3, 64, 3, 2, 96, 128, dt, names.data(), names.size(), fl.data(), by.data(), n) == n);
        CHECK(strlen(names.data()) > 100 && by[0] > 0 && fl[n - 2] > 0 && by[n - 1] > 0);   // (the last launch is the blob guard: no flops)
        CHECK(emavfi_forward_launches(3, 64, 3, 2, 96, 128, dt, names.data(), 40, fl.data(), by.data(), n) == EMAVFI_E_ARG);
        CHECK(emavfi_forward_launches(3, 64, 3, 2, 96, 128, dt, names.data(), names.size(), fl.data(), by.data(), n - 1) == EMAVFI_E_ARG);
        CHECK(emavfi_forward_launches(3, 64, 3, 2, 96, 128, dt, nullptr, 0, nullptr, by.data(), n) == n);
    }
    CHECK(emavfi_forward_launches(3, 8, 3, 1, 5, 7, EMAVFI_BF16, nullptr, 0, nullptr, nullptr, 0) > 0);
    CHECK(emavfi_forward_launches(3, 7, 3, 1, 64, 64, EMAVFI_BF16, nullptr, 0, nullptr, nullptr, 0) == EMAVFI_E_UNSUPPORTED);

    // the switch word: latched, settable, restored; the enumeration follows it
    const int old = emavfi_debug_switches(-1, 0);
    const int n0 = emavfi_forward_launches(3, 64, 3, 1, 64, 64, EMAVFI_BF16, nullptr, 0, nullptr, nullptr, 0);
    emavfi_debug_switches(-1, 1 | 2 | 4 | 8 | 64);
    CHECK(emavfi_forward_launches(3, 64, 3, 1, 64, 64, EMAVFI_BF16, nullptr, 0, nullptr, nullptr, 0) > n0);
    emavfi_debug_switches(0, old);
    CHECK(emavfi_debug_switches(-1, 0) == old);
    CHECK(emavfi_layout_tag() >= 0 && emavfi_layout_tag() < 64);

    // argument guards of the entries that would launch: refused on the host, nothing dereferenced
    void *fake = (void *)(uintptr_t)256;
    const float *ff = (const float *)fake;
    float *fo = (float *)fake;
    CHECK(emavfi_forward(3, 64, 3, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 1, 8, 8, EMAVFI_F32, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward(3, 64, 3, fake, 1000, ff, ff, fo, fake, (size_t)1 << 40, 1, 64, 64, EMAVFI_BF16, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(strstr(emavfi_last_error(), "packed blob has 1000 bytes") != nullptr);
    CHECK(emavfi_forward(3, 64, 3, fake, (size_t)1 << 30, ff, ff, fo, fake, 16, 1, 64, 64, EMAVFI_BF16, nullptr, nullptr) == EMAVFI_E_WORKSPACE);
    CHECK(emavfi_forward(3, 64, 3, (void *)(uintptr_t)260, (size_t)1 << 30, ff, ff, fo, fake, (size_t)1 << 40, 1, 64, 64, EMAVFI_BF16, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward(3, 64, 3, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 1, 4096, 4096, EMAVFI_BF16, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward(3, 64, 3, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 1, 2880, 5120, EMAVFI_AMP16, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward(3, 64, 3, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 8, 8, EMAVFI_F32, nullptr, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward(3, 65, 3, fake, 0, ff, ff, fo, fake, 0, 1, 8, 8, EMAVFI_F32, nullptr, nullptr) == EMAVFI_E_UNSUPPORTED);
    void *evs[2] = {nullptr, nullptr};
    CHECK(emavfi_forward_profiled(3, 64, 3, fake, 0, ff, ff, fo, fake, 0, 1, 8, 8, EMAVFI_F32, nullptr, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward_profiled(3, 64, 3, nullptr, 0, ff, ff, fo, fake, 0, 1, 8, 8, EMAVFI_F32, evs, 2, nullptr) == EMAVFI_E_ARG);
    // the staged form (round 5): the same guards with stage / launch events present or absent
    CHECK(emavfi_forward_staged(3, 64, 3, nullptr, 0, ff, ff, fo, fake, 0, 1, 8, 8, EMAVFI_F32, nullptr, nullptr, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_forward_staged(3, 64, 3, fake, 1000, ff, ff, fo, fake, (size_t)1 << 40, 1, 64, 64, EMAVFI_BF16, evs, evs, 2, nullptr) == EMAVFI_E_ARG);
    CHECK(strstr(emavfi_last_error(), "packed blob has 1000 bytes") != nullptr);
    CHECK(emavfi_warp(nullptr, nullptr, nullptr, 1, 3, 8, 8, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_warp(ff, ff, fo, 0, 3, 8, 8, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_warp((const float *)(uintptr_t)260, ff, fo, 1, 3, 8, 8, nullptr) == EMAVFI_E_ARG);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd0[3] = {0.229f, 0.0f, 0.225f};
    CHECK(emavfi_preprocess_u8(nullptr, fo, 1, 8, 8, 3, mean, mean, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_preprocess_u8((const unsigned char *)fake, fo, 1, 8, 8, 3, mean, sd0, nullptr) == EMAVFI_E_ARG);   // std 0
    CHECK(emavfi_preprocess_u8((const unsigned char *)fake, fo, 1, 8, 8, 5, mean, mean, nullptr) == EMAVFI_E_ARG);
    const double dm[3] = {0.485, 0.456, 0.406};
    CHECK(emavfi_postprocess_u8(ff, nullptr, 1, 8, 8, 3, dm, dm, 1, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_conv3x3_workspace_bytes(1, 64, 64, 32, 32, 1, EMAVFI_BF16) > 0 && emavfi_conv3x3_workspace_bytes(1, 64, 64, 32, 32, 3, EMAVFI_F32) == 0);
    CHECK(emavfi_conv3x3_workspace_bytes(1, 100, 8, 8, 8, 1, EMAVFI_F32) == 0);
    CHECK(emavfi_conv3x3(nullptr, nullptr, nullptr, nullptr, 1, 3, 3, 8, 8, 1, 0, EMAVFI_F32, nullptr, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_conv3x3(ff, ff, ff, fo, 1, 64, 64, 32768, 32768, 1, 0, EMAVFI_BF16, fake, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_conv3x3(ff, ff, ff, fo, 1, 64, 64, 8, 8, 1, 9, EMAVFI_BF16, fake, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_conv3x3(ff, ff, ff, fo, 1, 64, 64, 8, 8, 1, 0, EMAVFI_BF16, fake, 16, nullptr) == EMAVFI_E_WORKSPACE);
    CHECK(emavfi_deform_conv2d_workspace_bytes(1, 67, 67, 16, 16, EMAVFI_BF16) > 0 && emavfi_deform_conv2d_workspace_bytes(1, 200, 8, 8, 8, EMAVFI_F32) == 0);
    CHECK(emavfi_deform_conv2d(ff, ff, ff, ff, ff, fo, 1, 67, 67, 16, 16, EMAVFI_BF16, fake, 16, nullptr) == EMAVFI_E_WORKSPACE);
    CHECK(emavfi_deform_conv2d(ff, ff, ff, ff, ff, fo, 1, 67, 67, 4096, 4096, EMAVFI_BF16, fake, 16, nullptr) == EMAVFI_E_ARG);
    for (int dt : dtypes) CHECK(emavfi_mdcn_workspace_bytes(2, 67, 75, 131, dt, 0) > 0);
    CHECK(emavfi_mdcn_workspace_bytes(2, 67, 75, 131, EMAVFI_F32X3, 0) == 0 && strstr(emavfi_last_error(), "F32X3"));
    CHECK(emavfi_context_workspace_bytes(2, 64, 75, 131, EMAVFI_F32X3) == 0 && emavfi_conv3x3_workspace_bytes(2, 67, 64, 75, 131, 1, EMAVFI_F32X3) > 0);
    // the split mode's pixels are two f16 halves: the bytes of the fp32 mode's activations, three weight copies per chunk
    CHECK(emavfi_conv3x3_workspace_bytes(2, 64, 64, 75, 131, 1, EMAVFI_F32X3) > emavfi_conv3x3_workspace_bytes(2, 64, 64, 75, 131, 1, EMAVFI_F16));
    CHECK(emavfi_mdcn_workspace_bytes(1, 67, 32, 32, EMAVFI_BF16, EMAVFI_MDCN_SPLIT_TAIL | EMAVFI_MDCN_IN_F16 | EMAVFI_MDCN_OUT_F16) > 0);
    CHECK(emavfi_mdcn_workspace_bytes(1, 66, 32, 32, EMAVFI_F32, 0) == 0 && emavfi_mdcn_workspace_bytes(1, 3, 32, 32, EMAVFI_F32, 0) == 0);
    CHECK(emavfi_mdcn_workspace_bytes(1, 67, 32, 32, EMAVFI_F32, EMAVFI_MDCN_IN_F16) == 0 && emavfi_mdcn_workspace_bytes(1, 67, 32, 32, EMAVFI_BF16, 8) == 0);
    CHECK(emavfi_mdcn(nullptr, ff, ff, ff, ff, fo, 1, 67, 8, 8, EMAVFI_BF16, 0, fake, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_mdcn(ff, ff, ff, ff, nullptr, fo, 1, 67, 8, 8, EMAVFI_BF16, 0, fake, 16, nullptr) == EMAVFI_E_WORKSPACE);
    CHECK(emavfi_mdcn_profiled(ff, ff, ff, ff, ff, fo, 1, 67, 8, 8, EMAVFI_BF16, 0, fake, 16, nullptr, 0, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_mdcn_profiled(ff, ff, ff, ff, nullptr, fo, 1, 67, 8, 8, EMAVFI_BF16, 0, fake, 16, evs, 2, nullptr) == EMAVFI_E_WORKSPACE);
    for (int dt : dtypes) CHECK(emavfi_context_workspace_bytes(2, 64, 75, 131, dt) > 0 && emavfi_reconstruct_workspace_bytes(2, 8, 23, 37, dt) > 0);
    CHECK(emavfi_context_workspace_bytes(1, 7, 32, 32, EMAVFI_BF16) == 0 && emavfi_reconstruct_workspace_bytes(0, 64, 32, 32, EMAVFI_BF16) == 0);
    {
        const float *eight[8] = {ff, ff, ff, ff, ff, ff, ff, nullptr};
        CHECK(emavfi_context(ff, eight, fo, 1, 64, 16, 16, EMAVFI_BF16, fake, (size_t)1 << 40, nullptr) == EMAVFI_E_ARG && strstr(emavfi_last_error(), "params[7]"));
        CHECK(emavfi_context(nullptr, eight, fo, 1, 64, 16, 16, EMAVFI_BF16, fake, 0, nullptr) == EMAVFI_E_ARG);
        const float *six[6] = {ff, ff, ff, ff, ff, ff};
        CHECK(emavfi_reconstruct(ff, six, fo, 1, 64, 16, 16, EMAVFI_BF16, fake, 16, nullptr) == EMAVFI_E_WORKSPACE);
        CHECK(emavfi_reconstruct(ff, six, fo, 1, 64, 4096, 4096, EMAVFI_BF16, fake, 16, nullptr) == EMAVFI_E_ARG);
    }
    CHECK(emavfi_pack_weights(3, 64, 3, nullptr, 40, fake, 1 << 20, EMAVFI_BF16, nullptr) == EMAVFI_E_ARG);
    std::vector<const void *> params(40, fake);
    CHECK(emavfi_pack_weights(3, 64, 3, params.data(), 39, fake, (size_t)1 << 30, EMAVFI_BF16, nullptr) == EMAVFI_E_ARG);
    CHECK(emavfi_pack_weights(3, 64, 3, params.data(), 40, fake, 1000, EMAVFI_BF16, nullptr) == EMAVFI_E_WORKSPACE);
    params[7] = nullptr;
    CHECK(emavfi_pack_weights(3, 64, 3, params.data(), 40, fake, (size_t)1 << 30, EMAVFI_BF16, nullptr) == EMAVFI_E_ARG);

    // the blob header check on host memory
    const uint32_t tag = (uint32_t)emavfi_layout_tag();
    for (int mid : {8, 64}) {
        std::vector<unsigned char> good = host_blob(mid, EMAVFI_BF16, tag, EMAVFI_VERSION);
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_BF16, good.data(), good.size()) == EMAVFI_OK);
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_F16, good.data(), good.size()) == EMAVFI_E_ARG);
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_BF16, good.data(), good.size() - 1) == EMAVFI_E_ARG);
        std::vector<unsigned char> bad = good;
        bad[bad.size() - 3] ^= 0x40;
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_BF16, bad.data(), bad.size()) == EMAVFI_E_ARG && strstr(emavfi_last_error(), "checksum"));
        std::vector<unsigned char> other = host_blob(mid, EMAVFI_BF16, tag ^ 2u, EMAVFI_VERSION);
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_BF16, other.data(), other.size()) == EMAVFI_E_ARG && strstr(emavfi_last_error(), "layout switches"));
        std::vector<unsigned char> oldv = host_blob(mid, EMAVFI_BF16, tag, 300);
        CHECK(emavfi_packed_check(3, mid, 3, EMAVFI_BF16, oldv.data(), oldv.size()) == EMAVFI_E_ARG && strstr(emavfi_last_error(), "version"));
    }
    CHECK(emavfi_packed_check(3, 8, 3, EMAVFI_BF16, nullptr, 100) == EMAVFI_E_ARG);

    if (g_fail) { fprintf(stderr, "host_check: %d check(s) failed\n", g_fail); return 1; }
    printf("host_check: ok\n");
    return 0;
}
