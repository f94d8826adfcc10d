// kernels.h — host-callable launchers of the gfx950 kernels of libbatchsim.so.
//
// Every kernel family is its own translation unit (k_serial.hip, k_narrow.hip, and one object per
// tree-parallel instantiation of physics_tp.h::tp_run built from k_tp_kernels.hip with -DTPK_<name>),
// so the library builds in parallel; batchsim.hip (the C-ABI) only sees these launchers.
#pragma once
#include "physics_dev.h"
#include "physics_tp.h"
#include "physics_contact.h"

#define BS_BLOCK 64
#define NB_ENVS 16     // environments per block of k_c_narrow_flat / k_c_narrow_ws
#define NB_THREADS 128
#define NW_THREADS 128
#define NL_THREADS 64  // one wavefront per block measured best (3.21 / 3.25 / 3.30 ms per step for 64 / 128 / 256)
#ifndef NL_OCC
#define NL_OCC 2        // wavefronts per SIMD k_c_narrow_list is compiled for (register budget 512 / NL_OCC)
#endif
#ifndef NL_MIN_SHARE
#define NL_MIN_SHARE 64   // least number of work-list items a wavefront of k_c_narrow_list takes: one per lane.  Only small
                          // batches come down to it (a share is total / grid otherwise); 192 made their few wavefronts
                          // work through three items per lane in sequence: 4096 envs 7.9 -> 10.6 M env-steps/s
#endif
#ifndef TPC_OCC
#define TPC_OCC 2
#endif

// ---- serial layout (one environment per lane), k_serial.hip
void bs_launch_step(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int nsteps);
void bs_launch_forward(dim3 g, hipStream_t st, const DevModel& m, const DevData& d);
void bs_launch_stage(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int stage);
void bs_launch_reset(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const uint8_t* mask);
void bs_launch_check_state(dim3 g, hipStream_t st, const DevModel& m, const DevData& d);
void bs_launch_rk4_advance(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int i);
void bs_launch_frames(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, double* xmat, double* gpos,
                      double* gmat);
void bs_launch_pack_obs(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, float* out, long stride, int nq);
void bs_launch_env_pre(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const double* action, long ld);
void bs_launch_env_post(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const double* action, long ld, int* t,
                        int episode_steps, float* obs, float* term_obs, long stride, float* reward, uint8_t* flags, int nq);
void bs_launch_c_pos(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const ColData& c, int check);
void bs_launch_c_broad(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c);
void bs_launch_c_dyn(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const ColModel& cm, const DevData& d,
                     const ColData& c, const ConData& k, int integrate, int lds_hessian);
hipError_t bs_c_dyn_set_lds(int bytes);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) on k_c_dyn

// ---- narrowphase, k_narrow.hip
void bs_launch_narrow_flat(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c, int prim);
void bs_launch_narrow_ws(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c);
void bs_launch_narrow_list(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c, int prim);
#ifdef NP_PROFILE
void bs_np_profile_dump();
#endif

// ---- tree-parallel layout (one lane per body), k_tp_kernels.hip; grid = ceil(n_envs / 2) one-wavefront blocks
#define BS_TP_ARGS dim3 g, size_t lds, hipStream_t st, const DevModel &m, const TpModel &t, const DevData &d, int nsteps, int mode, int dump
void bs_launch_tp(BS_TP_ARGS);          // Newton, Euler
void bs_launch_tp_chain(BS_TP_ARGS);          // Newton, Euler
void bs_launch_tp_chain4(BS_TP_ARGS);         // ... hand-shaped models (TpModel::quad_top)
void bs_launch_tp_cg(BS_TP_ARGS);       // CG, Euler
void bs_launch_tp_pgs(BS_TP_ARGS);      // PGS, Euler (lds: M^-1 of the two environments)
void bs_launch_tp_rk4(BS_TP_ARGS);
void bs_launch_tp_rk4_cg(BS_TP_ARGS);
void bs_launch_tp_rk4_pgs(BS_TP_ARGS);
void bs_launch_tp_full(BS_TP_ARGS);     // bs_forward: every intermediate written
void bs_launch_tp_full_cg(BS_TP_ARGS);
void bs_launch_tp_full_pgs(BS_TP_ARGS);
void bs_launch_tp_col(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const TpModel& t, const DevData& d,
                      const ColModel& cm, const ColData& cd);
#define BS_TPC_ARGS dim3 g, size_t lds, hipStream_t st, const DevModel &m, const TpModel &t, const DevData &d, const ColModel &cm, const ColData &cd, int mode, int dump
void bs_launch_tpc(BS_TPC_ARGS);        // contacts, Newton
void bs_launch_tpc_chain(BS_TPC_ARGS);        // contacts, Newton
void bs_launch_tpc_cg(BS_TPC_ARGS);     // contacts, CG
void bs_launch_tpc_pgs(BS_TPC_ARGS);    // contacts, PGS
void bs_launch_tpc_full(BS_TPC_ARGS);     // bs_forward with contacts, Newton: every intermediate written
void bs_launch_tpc_full_cg(BS_TPC_ARGS);  // ... CG
void bs_launch_tpc_full_pgs(BS_TPC_ARGS); // ... PGS
// ---- the WIDE build of the same kernels (k_tp_kernels.hip compiled with -DTP_LPE=64): one environment per
// wavefront, nbody <= 63, up to 64 contacts per environment; grid = n_envs one-wavefront blocks.  No CHAIN instantiations.
void bs_launch_tp_w(BS_TP_ARGS);
void bs_launch_tp_cg_w(BS_TP_ARGS);
void bs_launch_tp_pgs_w(BS_TP_ARGS);
void bs_launch_tp_rk4_w(BS_TP_ARGS);
void bs_launch_tp_rk4_cg_w(BS_TP_ARGS);
void bs_launch_tp_rk4_pgs_w(BS_TP_ARGS);
void bs_launch_tp_full_w(BS_TP_ARGS);
void bs_launch_tp_full_cg_w(BS_TP_ARGS);
void bs_launch_tp_full_pgs_w(BS_TP_ARGS);
void bs_launch_tp_col_w(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const TpModel& t, const DevData& d,
                        const ColModel& cm, const ColData& cd);
void bs_launch_tpc_w(BS_TPC_ARGS);
void bs_launch_tpc_cg_w(BS_TPC_ARGS);
void bs_launch_tpc_pgs_w(BS_TPC_ARGS);
void bs_launch_tpc_full_w(BS_TPC_ARGS);
void bs_launch_tpc_full_cg_w(BS_TPC_ARGS);
void bs_launch_tpc_full_pgs_w(BS_TPC_ARGS);
