#include "api.h"
#include "cputime.h"
#include "fatal.h"

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#ifndef TGSF_LIB_REL
#define TGSF_LIB_REL "../libtgsf.so"        // relative to the executable (tgsfilter_amd/bin/tgsfilter)
#endif
// What tgsf_backend() of the library must begin with.  The product runs on the HIP build only; the test builds of this
// program (make emul / asan / tsan) are compiled to expect the emulation instead -- a property of the executable, not of
// the environment: TGSF_LIB says where the library is, never what kind of library is acceptable.
#ifndef TGSF_EXPECT_BACKEND
#define TGSF_EXPECT_BACKEND "hip"
#endif

namespace host {

namespace {
Api g_api;
std::thread g_thread;
std::string g_error;
double g_load_s = 0, g_device_s = 0;
bool g_started = false;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string lib_path()
{
    if (const char* e = getenv("TGSF_LIB")) return e;
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    std::string dir = n > 0 ? std::string(exe, (size_t)n) : std::string(".");
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    return dir + "/" + TGSF_LIB_REL;
}

void load(std::vector<int> devices)
{
    CpuScope cpu(CPU_LOADER);
    const double t0 = now_s();
    const std::string path = lib_path();
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { g_error = std::string("cannot load ") + path + ": " + dlerror() + " (there is no CPU fallback)"; return; }
#define BIND(field, sym) \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, sym)); \
    if (!g_api.field) { g_error = std::string("symbol ") + sym + " missing in " + path; return; }
    BIND(abi_version, "tgsf_abi_version")
    BIND(backend, "tgsf_backend")
    BIND(prepare_device, "tgsf_prepare_device")
    BIND(create, "tgsf_create")
    BIND(destroy, "tgsf_destroy")
    BIND(submit, "tgsf_submit")
    BIND(wait, "tgsf_wait")
    BIND(counters_len, "tgsf_counters_len")
    BIND(counters, "tgsf_counters")
    BIND(counters_used, "tgsf_counters_used")
    BIND(align_windows, "tgsf_align_windows")
    BIND(last_error, "tgsf_last_error")
    BIND(profile, "tgsf_profile")
    BIND(stage_times, "tgsf_stage_times")
    BIND(stage_name, "tgsf_stage_name")
    BIND(device_location, "tgsf_device_location")
    BIND(counters_merge, "tgsf_counters_merge")
#undef BIND
    if (g_api.abi_version() != TGSF_ABI_VERSION) { g_error = path + ": ABI version mismatch"; return; }
    {
        const char* b = g_api.backend();
        const std::string want = TGSF_EXPECT_BACKEND;
        if (!b || std::string(b).compare(0, want.size(), want) != 0) {
            g_error = path + " is the \"" + (b ? b : "?") + "\" build of the library; this program runs on the \"" + want +
                      "\" build only (there is no CPU fallback; is TGSF_LIB set by accident?)";
            return;
        }
    }
    g_load_s = now_s() - t0;
    // every device on a thread of its own (a device takes 0.1-0.3 s to come up); failures surface at tgsf_create, with
    // its message
    std::vector<std::thread> up;
    for (size_t i = 1; i < devices.size(); i++) {
        bool seen = false;
        for (size_t j = 0; j < i; j++) seen = seen || devices[j] == devices[i];
        if (!seen) up.emplace_back([d = devices[i]] { CpuScope c2(CPU_LOADER); (void)g_api.prepare_device(d); });
    }
    if (!devices.empty()) (void)g_api.prepare_device(devices[0]);
    for (std::thread& t : up) t.join();
    g_device_s = now_s() - t0 - g_load_s;
}
}  // namespace

void lib_start(const std::vector<int>& devices)
{
    g_started = true;
    g_thread = std::thread(load, devices);
}

const Api& lib()
{
    if (!g_started) lib_start({});
    if (g_thread.joinable()) g_thread.join();
    if (!g_error.empty()) {
        std::cerr << "Error: " << g_error << std::endl;
        quit(255);
    }
    return g_api;
}

namespace {
RcclApi g_rccl;
bool g_rccl_ok = false, g_rccl_started = false;
std::thread g_rccl_thread;

void load_rccl()
{
    std::string path = lib_path();
    const size_t slash = path.rfind('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libtgsf_rccl.so";
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);      // (brings librccl in: a few tenths of a second)
    if (!h) return;
#define BIND(field, sym) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym)); if (!g_rccl.field) return;
    BIND(unique_id, "tgsf_rccl_unique_id")
    BIND(comm_init, "tgsf_rccl_comm_init")
    BIND(comm_count, "tgsf_rccl_comm_count")
    BIND(comm_destroy, "tgsf_rccl_comm_destroy")
    BIND(allreduce_counters, "tgsf_rccl_allreduce_counters")
    BIND(last_error, "tgsf_rccl_last_error")
#undef BIND
    g_rccl_ok = true;
}
}  // namespace

void rccl_start()
{
    if (g_rccl_started) return;
    g_rccl_started = true;
    g_rccl_thread = std::thread(load_rccl);
}

const RcclApi* rccl_lib()
{
    rccl_start();
    if (g_rccl_thread.joinable()) g_rccl_thread.join();
    return g_rccl_ok ? &g_rccl : nullptr;
}

void lib_times(double& load_s, double& device_s) { load_s = g_load_s; device_s = g_device_s; }

}  // namespace host
