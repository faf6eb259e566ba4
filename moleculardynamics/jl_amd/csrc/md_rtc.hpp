// md_rtc.hpp -- run-time compilation of the force kernels around a user-supplied potential
// (md_set_potential_source).  This is the GPU form of the reference's plugin API: a Potential
// subtype overloads evaluate(pot, r, sigma1, sigma2) -> (u, f) (src/types.jl:1-6,
// src/pairwise.jl:31, README.md:86-145); here it supplies the same function as HIP source
//     __device__ void <entry>(double r, double s1, double s2, const double* p, double* u, double* f);
// and hiprtc compiles it into the very kernels the built-in potentials use (POT_CUSTOM branch
// of pair_eval), so lists, tiling, masking and reductions are shared.
#pragma once
#include <hip/hiprtc.h>

#include "md_kernels_src.inc"

struct RtcModule {
    hipModule_t module = nullptr;
    // [dim-2][want_uw][kick]
    hipFunction_t tile[2][2][2] = {};
    hipFunction_t global[2][2][2] = {};
    std::string log;
    ~RtcModule()
    {
        if (module) (void)hipModuleUnload(module);
    }
};

inline void rtc_check(hiprtcResult r, const char *what, hiprtcProgram prog = nullptr)
{
    if (r == HIPRTC_SUCCESS) return;
    std::string msg = std::string(what) + ": " + hiprtcGetErrorString(r);
    if (prog) {
        size_t n = 0;
        if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
            std::string log(n, '\0');
            (void)hiprtcGetProgramLog(prog, &log[0]);
            msg += "\n" + log;
        }
    }
    throw std::runtime_error(msg);
}

inline RtcModule *rtc_build(const char *user_src, const char *entry)
{
    std::string src;
    src += "#define MD_RTC 1\n#define MD_HAVE_USER_POTENTIAL 1\n#define MD_USER_ENTRY ";
    src += entry;
    src += "\n";
    for (int i = 0; md_kernels_src_pieces[i]; ++i) src += md_kernels_src_pieces[i];
    src += "\n// ---- user potential ----\n";
    src += user_src;
    src += "\n";
    hiprtcProgram prog;
    rtc_check(hiprtcCreateProgram(&prog, src.c_str(), "md_user_potential.hip", 0, nullptr, nullptr), "hiprtcCreateProgram");
    std::vector<std::string> names;
    for (int d = 2; d <= 3; ++d)
        for (int uw = 0; uw < 2; ++uw)
            for (int kk = 0; kk < 2; ++kk) {
                char b[160];
                snprintf(b, sizeof b, "k_force_tile<%d, %d, false, %s, %s, false>", d, POT_CUSTOM, uw ? "true" : "false", kk ? "true" : "false");
                names.push_back(b);
                snprintf(b, sizeof b, "k_force<%d, %d, false, %s, %s>", d, POT_CUSTOM, uw ? "true" : "false", kk ? "true" : "false");
                names.push_back(b);
            }
    for (auto &nm : names) rtc_check(hiprtcAddNameExpression(prog, nm.c_str()), "hiprtcAddNameExpression", prog);
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult cr = hiprtcCompileProgram(prog, 3, opts);
    if (cr != HIPRTC_SUCCESS) {
        std::string keep;
        try {
            rtc_check(cr, "compiling the user potential failed", prog);
        } catch (const std::exception &e) {
            keep = e.what();
        }
        (void)hiprtcDestroyProgram(&prog);
        throw std::runtime_error(keep);
    }
    size_t csize = 0;
    rtc_check(hiprtcGetCodeSize(prog, &csize), "hiprtcGetCodeSize", prog);
    std::vector<char> code(csize);
    rtc_check(hiprtcGetCode(prog, code.data()), "hiprtcGetCode", prog);
    RtcModule *m = new RtcModule();
    try {
        if (hipModuleLoadData(&m->module, code.data()) != hipSuccess) throw std::runtime_error("hipModuleLoadData failed");
        size_t idx = 0;
        for (int d = 2; d <= 3; ++d)
            for (int uw = 0; uw < 2; ++uw)
                for (int kk = 0; kk < 2; ++kk) {
                    const char *low = nullptr;
                    rtc_check(hiprtcGetLoweredName(prog, names[idx].c_str(), &low), "hiprtcGetLoweredName", prog);
                    if (hipModuleGetFunction(&m->tile[d - 2][uw][kk], m->module, low) != hipSuccess)
                        throw std::runtime_error("hipModuleGetFunction failed for the tiled force kernel");
                    ++idx;
                    rtc_check(hiprtcGetLoweredName(prog, names[idx].c_str(), &low), "hiprtcGetLoweredName", prog);
                    if (hipModuleGetFunction(&m->global[d - 2][uw][kk], m->module, low) != hipSuccess)
                        throw std::runtime_error("hipModuleGetFunction failed for the force kernel");
                    ++idx;
                }
    } catch (...) {
        delete m;
        (void)hiprtcDestroyProgram(&prog);
        throw;
    }
    (void)hiprtcDestroyProgram(&prog);
    return m;
}
