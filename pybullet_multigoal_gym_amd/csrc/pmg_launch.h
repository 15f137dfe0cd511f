/* pmg_launch.h -- host-callable launchers of the kernels in pmg_kernels.hip */
#ifndef PMG_LAUNCH_H
#define PMG_LAUNCH_H
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

hipError_t pmg_launch_plan(const pmg::EnvParams& P, const float* d_actions, hipStream_t s);
/* packed != 0 selects the fast paths: four envs per wavefront (reach, one-object tasks; pmg_packed.h) and the
 * small-contact-store list of the multi-block tasks, which runs on `side` concurrently with the full-store list */
hipError_t pmg_launch_step(const pmg::EnvParams& P, const float* d_actions, hipStream_t s, int packed, hipStream_t side,
                           hipEvent_t ev_fork, hipEvent_t ev_join);
hipError_t pmg_launch_reset(const pmg::EnvParams& P, const unsigned char* d_mask, hipStream_t s, int done_only = 0);
hipError_t pmg_launch_sub_goal(const pmg::EnvParams& P, const unsigned char* d_mask, int level, hipStream_t s);
hipError_t pmg_launch_reward(const float* ag, const float* dg, long long B, int G, float thr, int binary, float* reward,
                             unsigned char* ok, hipStream_t s);
#endif
