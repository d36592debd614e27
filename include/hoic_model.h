/* hoic_model.h — wire format of the compiled model ("model blob").
 *
 * Replaces, for the HOIC hand+object model family, what the reference obtains from
 * mujoco_py.load_model_from_path() (uhc/khrylib/rl/envs/common/mujoco_env.py:18-34) after the
 * MJCF merge (uhc/data_loaders/mjxml/MujocoXML.py:72-106).  Produced by hoic_amd/mjcf.py.
 *
 * Layout (little endian):
 *   header   : char magic[8] = "HOICMDL1"; int32 version (=1); int32 nentries;
 *   table    : nentries x hoic_blob_entry  (sorted by name)
 *   payload  : each array 64-byte aligned at entry.offset (bytes from blob start)
 * Arrays are float64 (dtype 0) or int32 (dtype 1), C-contiguous, up to 4 dims.
 * Names are the MuJoCo mjModel field names where one exists (body_pos, jnt_axis, geom_size ...),
 * plus pair_* (static collision pair list with mixed contact parameters) and the env-glue
 * indices the reference derives from names (uhc/envs/ho_im4.py:74-97).
 */
#ifndef HOIC_MODEL_H
#define HOIC_MODEL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOIC_BLOB_MAGIC "HOICMDL1"

typedef struct hoic_blob_header {
  char magic[8];
  int32_t version;
  int32_t nentries;
} hoic_blob_header;

typedef struct hoic_blob_entry {
  char name[32];
  int32_t dtype; /* 0 = float64, 1 = int32 */
  int32_t ndim;
  int32_t shape[4];
  int64_t offset;
  int64_t nbytes;
} hoic_blob_entry;

/* MuJoCo enum values used by the tables */
enum { HOIC_JNT_FREE = 0, HOIC_JNT_BALL = 1, HOIC_JNT_SLIDE = 2, HOIC_JNT_HINGE = 3 };
enum { HOIC_GEOM_PLANE = 0, HOIC_GEOM_SPHERE = 2, HOIC_GEOM_CAPSULE = 3, HOIC_GEOM_BOX = 6, HOIC_GEOM_MESH = 7 };

/* capacities both implementations size their fixed tables with */
enum {
  HOIC_MAX_BODY = 28,
  HOIC_MAX_JNT = 28,
  HOIC_MAX_NQ = 33,
  HOIC_MAX_NV = 32,
  HOIC_MAX_NU = 26,
  HOIC_MAX_GEOM = 28,
  HOIC_MAX_PAIR = 128,
  HOIC_MAX_MESH = 4,
  HOIC_MAX_MESHVERT = 2048,  /* all hull vertices of a model's meshes (banana: 231 + 707 + 939) */
  HOIC_MAX_MESHPLANE = 4096, /* hull face planes n.x <= d, mesh frame */
  /* The hull tables are ordered (hoic_amd/mjcf.py coherent_order) so that consecutive runs of this many vertices are
     spatially compact and consecutive runs of this many faces have similar normals: the simulator bounds each run at load
     time and skips the runs a query cannot touch.  Results never depend on the skipping (exact bounds). */
  HOIC_HULL_RUN_VERTS = 64,
  HOIC_HULL_RUN_FACES = 32,
  HOIC_OBS_DIM = 617,  /* get_full_obs_v5(w=5), uhc/envs/ho_im4.py:280-356 */
  HOIC_ACT_DIM = 32,   /* 26 PD targets + 3 residual force + 3 residual torque, ho_im4.py:145-156 */
  HOIC_NHANDBODY = 21, /* bodies named link*, ho_im4.py:77-78 */
  HOIC_NREWARD_INFO = 9
};

#ifdef __cplusplus
}
#endif
#endif
