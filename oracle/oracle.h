/* oracle.h — CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in MuJoCo 2.1.0
 * (closed binary fetched by reference install.sh:4-5, bound by mujoco-py,
 * requirements.txt:1; call sites src/Mujoco_XML.py:4,341-345).  Neither is present in
 * /root/reference and the reference holds no tests or golden vectors,
 * so this file restates MuJoCo's published algorithm ("Computation" chapter) and is
 * pinned only by the structural facts of output/DexterousDynamos.xml and by
 * identity / brute-force tests in tests/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library — never the product path (mujoco_amd/).
 *
 * One environment, AoS, fp64, dense constraint Jacobian: the layout MuJoCo itself uses.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model {
  /* sizes */
  int32_t nbody, njnt, nv, nu, ngeom, neq, nmesh, nM, npair, nhullvert, nhulladj;
  /* options (MuJoCo <option> defaults [EXT] unless the XML overrides them) */
  int32_t integrator, solver, iterations, ls_iterations, disableflags, mpr_iterations;
  double timestep, gravity[3], tolerance, ls_tolerance, impratio, mpr_tolerance, meaninertia;
  /* bodies */
  const int32_t *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum,
      *body_dofadr, *body_dofnum;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass,
      *body_inertia, *body_invweight0;
  /* joints */
  const int32_t *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_stiffness, *jnt_solref,
      *jnt_solimp, *qpos0, *qpos_spring;
  /* dofs */
  const int32_t *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp,
      *dof_invweight0;
  /* geoms */
  const int32_t *geom_type, *geom_bodyid, *geom_dataid, *geom_condim, *geom_priority;
  const double *geom_pos, *geom_quat, *geom_size, *geom_friction, *geom_margin, *geom_gap,
      *geom_solref, *geom_solimp, *geom_solmix, *geom_rbound;
  /* convex hulls (mesh-local principal frame) */
  const int32_t *mesh_vertadr, *mesh_vertnum, *hull_adj_start, *hull_adj;
  const double *hull_vert;
  /* actuators (joint transmission) */
  const int32_t *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited;
  const double *actuator_ctrlrange, *actuator_forcerange, *actuator_gear, *actuator_gainprm,
      *actuator_biasprm;
  /* candidate collision pairs (compile-time filtered) */
  const int32_t *pair_geom1, *pair_geom2;
  /* joint equalities */
  const int32_t *eq_obj1id, *eq_obj2id, *eq_active;
  const double *eq_data, *eq_solref, *eq_solimp;
} orc_model;

typedef struct orc_contact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5];
  int32_t dim, geom1, geom2, efc_address;
} orc_contact;

#define ORC_MAXCON 128

typedef struct orc_data {
  /* state */
  double time;
  double *qpos, *qvel, *ctrl, *qfrc_applied, *qacc, *qacc_warmstart;
  /* position stage */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat;
  double *subtree_com, *cinert, *cdof, *crb, *qM, *qLD, *qLDiagInv;
  /* velocity / force stage */
  double *cvel, *cdof_dot, *qfrc_bias, *qfrc_passive, *actuator_force, *qfrc_actuator,
      *qfrc_smooth, *qacc_smooth, *qfrc_constraint;
  /* constraints */
  int32_t nefc, nefc_max, ncon, ne, nf, nl;
  int32_t *efc_type, *efc_id, *efc_state;
  double *efc_J, *efc_pos, *efc_margin, *efc_frictionloss, *efc_diagApprox, *efc_KBIP, *efc_D,
      *efc_R, *efc_vel, *efc_aref, *efc_force;
  orc_contact contact[ORC_MAXCON];
  int32_t solver_iter;
  int32_t warning; /* bit 0: bad qpos, 1: bad qvel, 2: bad qacc, 3: contact overflow */
  /* narrowphase statistics (for the work model of SURVEY.md §8d) */
  int64_t stat_npair_narrow, stat_mpr_iter;
  /* private scratch */
  double *scratch;
} orc_data;

/* Deep-copies every array: the caller's buffers may be released afterwards. */
orc_model *orc_model_create(const orc_model *desc);
void orc_model_destroy(orc_model *m);
orc_data *orc_data_create(const orc_model *m);
void orc_data_destroy(orc_data *d);
void orc_reset(const orc_model *m, orc_data *d);

/* pipeline stages (names follow the MuJoCo routines they restate) */
void orc_kinematics(const orc_model *m, orc_data *d);      /* A1 */
void orc_com_pos(const orc_model *m, orc_data *d);         /* A2 */
void orc_crb(const orc_model *m, orc_data *d);             /* A3 */
void orc_factor_m(const orc_model *m, orc_data *d);        /* A4 */
void orc_solve_m(const orc_model *m, const orc_data *d, double *x);
void orc_mul_m(const orc_model *m, const orc_data *d, double *res, const double *v);
void orc_full_m(const orc_model *m, const orc_data *d, double *dense);
void orc_collision(const orc_model *m, orc_data *d);       /* A5 + A6 */
void orc_make_constraint(const orc_model *m, orc_data *d); /* A7 */
void orc_fwd_velocity(const orc_model *m, orc_data *d);    /* A8 */
void orc_fwd_actuation(const orc_model *m, orc_data *d);   /* A9 */
void orc_fwd_acceleration(const orc_model *m, orc_data *d);/* A10 */
void orc_fwd_constraint(const orc_model *m, orc_data *d);  /* A11 */
void orc_forward(const orc_model *m, orc_data *d);
void orc_step(const orc_model *m, orc_data *d);            /* A12 / A13 / A14 */
/* rne with optional acceleration term, exposed for identity tests */
void orc_rne(const orc_model *m, orc_data *d, int flg_acc, double *result);
/* total constraint+Gauss cost at a given qacc (for KKT / optimiser cross-checks) */
double orc_cost(const orc_model *m, orc_data *d, const double *qacc);
/* single pair narrowphase exposed for brute-force geometric tests:
   returns number of contacts (0/1), writes dist, pos[3], normal[3] */
int orc_narrow_pair(const orc_model *m, const orc_data *d, int g1, int g2, double margin,
                    double *dist, double *pos, double *normal);

#ifdef __cplusplus
}
#endif
#endif
