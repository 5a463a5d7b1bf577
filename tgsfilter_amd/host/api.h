// api.h -- the C ABI of include/tgsf.h, bound at run time.  The command line does not link libtgsf.so:
// loading the HIP runtime's libraries costs a few tenths of a second before main() would even start, so a
// helper thread dlopen()s the library (and brings the device up, tgsf_prepare_device) while the main thread
// already maps and indexes the input.  There is no other implementation behind these pointers: if the
// library or a device is missing the program stops with an error.
#pragma once
#include <vector>

#include "tgsf.h"
#include "tgsf_rccl.h"

namespace host {

struct Api {
    decltype(&tgsf_abi_version) abi_version;
    decltype(&tgsf_backend) backend;
    decltype(&tgsf_prepare_device) prepare_device;
    decltype(&tgsf_create) create;
    decltype(&tgsf_destroy) destroy;
    decltype(&tgsf_submit) submit;
    decltype(&tgsf_wait) wait;
    decltype(&tgsf_counters_len) counters_len;
    decltype(&tgsf_counters) counters;
    decltype(&tgsf_counters_used) counters_used;
    decltype(&tgsf_align_windows) align_windows;
    decltype(&tgsf_last_error) last_error;
    decltype(&tgsf_profile) profile;              // stage timing (TGSF_TIMING only)
    decltype(&tgsf_stage_times) stage_times;
    decltype(&tgsf_stage_name) stage_name;
    decltype(&tgsf_device_location) device_location;
    decltype(&tgsf_counters_merge) counters_merge;
};

// include/tgsf_rccl.h (libtgsf_rccl.so, beside libtgsf.so): the tally all-reduce of a job of several rank processes
struct RcclApi {
    decltype(&tgsf_rccl_unique_id) unique_id;
    decltype(&tgsf_rccl_comm_init) comm_init;
    decltype(&tgsf_rccl_comm_count) comm_count;
    decltype(&tgsf_rccl_comm_destroy) comm_destroy;
    decltype(&tgsf_rccl_allreduce_counters) allreduce_counters;
    decltype(&tgsf_rccl_last_error) last_error;
};
// nullptr where the library is not installed or does not load (a box without RCCL): the caller then sums the tallies of
// its ranks over their sockets.  rccl_start() begins loading it on a thread of its own (it brings librccl in: a few
// tenths of a second, beside the pre-pass); rccl_lib() waits for that (and starts it if nobody has).
void rccl_start();
const RcclApi* rccl_lib();

// Starts the helper thread: dlopen + tgsf_prepare_device on each device.  Call once, early.
void lib_start(const std::vector<int>& devices);
// The bound entry points; blocks until the helper thread is done.  Exits with an error if loading failed.
const Api& lib();
// seconds the helper thread took (library load, device bring-up), for TGSF_TIMING
void lib_times(double& load_s, double& device_s);

}  // namespace host
