/* loamx — C-ABI of the MI355X-native LOAM registration hot path.
 *
 * One opaque handle per reference class on the path; every entry point below is what a binding of that class
 * would call.  The reference interfaces replaced (all under /root/reference):
 *
 *   loamx_scanreg_*  <->  loam::BasicScanRegistration   include/loam_velodyne/BasicScanRegistration.h:135-163
 *                                                        (processScanlines src/lib/BasicScanRegistration.cpp:28-46,
 *                                                         extractFeatures :155-254, RegistrationParams .h:34-72)
 *   loamx_odom_*     <->  loam::BasicLaserOdometry      include/loam_velodyne/BasicLaserOdometry.h:16-48
 *                                                        (process src/lib/BasicLaserOdometry.cpp:196-666,
 *                                                         transformToEnd :57-87, updateIMU :181-194)
 *   loamx_map_*      <->  loam::BasicLaserMapping       include/loam_velodyne/BasicLaserMapping.h:80-111
 *                                                        (process src/lib/BasicLaserMapping.cpp:266-599,
 *                                                         optimizeTransformTobeMapped :626-926,
 *                                                         updateOdometry :607-621)
 *   loamx_tm_*       <->  loam::BasicTransformMaintenance include/loam_velodyne/BasicTransformMaintenance.h:44-66
 *   loamx_batch_*    new: the batched-sweep mode of BASELINE.json's north_star (independent sweeps against a frozen
 *                    shared map) — the unit that is sharded across GPUs.
 *
 * Conventions
 *  - Point clouds are caller-owned arrays of records with x,y,z float32 at byte offsets 0/4/8 and intensity float32
 *    at byte offset `intensity_offset` of each record, `stride` bytes apart.  pcl::PointXYZI is {stride 32,
 *    intensity_offset 16}; a packed float4 is {16, 12}.  Outputs are written with the same description.
 *  - Poses are float[6] = rot_x (pitch), rot_y (yaw), rot_z (roll), x, y, z in the LOAM camera frame, rotation order
 *    R = Ry*Rx*Rz (reference src/lib/math_utils.h:212-238).
 *  - Point coordinates handed to loamx_scanreg_process, loamx_odom_*, loamx_map_* and staged into a pipeline must be finite.  The
 *    reference's own pipeline guarantees that (MultiScanRegistration.cpp:187-191 drops non-finite returns — and so does
 *    loamx_scanreg_process_raw / loamx_pipeline_stage_step_raw); its pcl::removeNaNFromPointCloud calls in the odometry
 *    (BasicLaserOdometry.cpp:230, :252) are safeguards that never fire there.  Here a violation is an ERROR, not a silent drop: the
 *    call (for a pipeline: the step that first uses the sweep) returns LOAMX_E_INVALID — binned rings are checked on the device while
 *    the curvature pass reads them, feature clouds on the host while they are packed; full-resolution clouds that are only
 *    transformed (loamx_map_process full_res) are not checked.
 *  - Return value: 0 = processed, 1 = skipped (mirrors the reference's `false` / silent guards), < 0 = error;
 *    loamx_last_error() gives the text for the calling thread's last failing call.  No exception or abort crosses
 *    the ABI.
 *  - A handle is single-threaded (externally synchronised), owns its device memory and one HIP stream; calls are
 *    synchronous unless named *_async.  No pointer returned by the library outlives the next call on that handle.
 *  - The library needs a gfx950 GPU: creating a handle without one fails with LOAMX_E_NOGPU.  There is no CPU
 *    fallback anywhere in the library.
 */
#ifndef LOAMX_H
#define LOAMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOAMX_OK 0
#define LOAMX_SKIPPED 1
#define LOAMX_E_INVALID (-1)
#define LOAMX_E_CAPACITY (-2)
#define LOAMX_E_HIP (-3)
#define LOAMX_E_NOGPU (-4)
#define LOAMX_E_UNSUPPORTED (-5) /* an optional dependency was left out of this build (the multi-GPU exchanges without RCCL) */

/* caller-owned cloud description (input or output) */
typedef struct loamx_cloud {
  void* data;                /* first record */
  uint32_t count;            /* in: number of points (input) / capacity (output); out: points written */
  uint32_t stride;           /* bytes between records, >= 16 */
  uint32_t intensity_offset; /* byte offset of the float32 intensity inside a record (12 or 16) */
  uint32_t reserved;
} loamx_cloud;

const char* loamx_last_error(void);
/* number of visible HIP devices (0 if none / HIP unavailable); never fails */
int loamx_device_count(void);
/* ABI version of this header */
#define LOAMX_ABI_VERSION 6
int loamx_abi_version(void);
/* How this library was built, as "key=value;..." (static storage): abi=<n>; diag=0|1 (1: made with EXTRA=-DLOAMX_DIAG — the only kind of
 * build that reads the diagnostic LOAMX_* environment switches, some of which change results; a product library ignores them);
 * rccl=0|1; roctx=0|1.  A harness records it next to its measurements. */
const char* loamx_build_info(void);
/* Host memory pinned by the HIP runtime this library runs on (hipHostMalloc / hipHostFree; NULL when there is no device or no
 * memory): sweeps of packed {stride 16, intensity at 12} records handed over from such memory, and landing areas of that kind for
 * registered clouds, are copied by DMA from / to where they lie instead of through the handles' staging blocks.  Memory pinned by other
 * means through the same runtime (hipHostRegister, a framework's pinned allocator) is recognised as well; anything else works as before. */
void* loamx_host_alloc(size_t bytes);
void loamx_host_free(void* p);

/* ------------------------------------------------------------------------------------------------------------
 * Feature extraction  (BasicScanRegistration, IMU-less path)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_scanreg loamx_scanreg;

/* mirrors loam::RegistrationParams (BasicScanRegistration.h:34-72, defaults .h:37-44) */
typedef struct loamx_scanreg_config {
  float scan_period;                 /* 0.1 */
  int n_feature_regions;             /* 6 */
  int curvature_region;              /* 5 */
  int max_corner_sharp;              /* 2 */
  int max_surface_flat;              /* 4 */
  float less_flat_filter_size;       /* 0.2 */
  float surface_curvature_threshold; /* 0.1 */
  int device;                        /* HIP device ordinal */
  /* (ABI version 2) */
  int max_corner_less_sharp;         /* 20; must be >= max_corner_sharp (ScanRegistration.cpp:100-109).  0 = 10 x max_corner_sharp,
                                        what the RegistrationParams constructor derives (BasicScanRegistration.cpp:22) */
  int imu_history_size;              /* 200; >= 1 (ScanRegistration.cpp:59-66).  The reference's history buffer is created with 200
                                        entries and ensureCapacity only grows it (CircularBuffer.h:53-70), so values below 200
                                        behave as 200; at most 4096 here */
} loamx_scanreg_config;

void loamx_scanreg_default_config(loamx_scanreg_config* cfg);
loamx_scanreg* loamx_scanreg_create(const loamx_scanreg_config* cfg);
void loamx_scanreg_destroy(loamx_scanreg* h);
/* BasicScanRegistration::configure (BasicScanRegistration.cpp:49-53): new parameters for an existing handle; the IMU history,
 * the scan time and the sweep state are kept, as in the reference (cfg->device must be the handle's device) */
int loamx_scanreg_configure(loamx_scanreg* h, const loamx_scanreg_config* cfg);
/* processScanlines: `cloud` holds the rings concatenated in ring order, ring r occupying ring_size[r] points.
 * Outputs (any may be NULL): sharp, less_sharp, flat, less_flat; count fields return the sizes. */
int loamx_scanreg_process(loamx_scanreg* h, const loamx_cloud* cloud, const uint32_t* ring_size, uint32_t n_rings,
                          loamx_cloud* sharp, loamx_cloud* less_sharp, loamx_cloud* flat, loamx_cloud* less_flat);

/* Raw-sweep ingestion + feature extraction: loam::MultiScanRegistration::process(laserCloudIn, scanTime)
 * (src/lib/MultiScanRegistration.cpp:160-238) followed by processScanlines.  `mapper` mirrors loam::MultiScanMapper
 * (include/loam_velodyne/MultiScanRegistration.h:47-89; presets .h:60-75; validation .cpp:107-127).
 * raw_xyz: `count` records with x, y, z float32 at byte offsets 0/4/8 (sensor axes: x forward, y left, z up — the
 * /velodyne_points payload), `stride` bytes apart, in firing order.
 * Outputs (any may be NULL): `full` = the binned cloud in the LOAM frame, rings concatenated, intensity = ring + relTime
 * (laserCloud()); ring_size[n_scan_rings] = points per ring; then the four feature clouds as in loamx_scanreg_process. */
typedef struct loamx_multiscan_mapper {
  float lower_bound_deg;  /* vertical angle of the first ring */
  float upper_bound_deg;  /* vertical angle of the last ring */
  uint32_t n_scan_rings;
} loamx_multiscan_mapper;
int loamx_multiscan_mapper_preset(const char* sensor /* "VLP-16" | "HDL-32" | "HDL-64E" */, loamx_multiscan_mapper* out);
int loamx_scanreg_process_raw(loamx_scanreg* h, const loamx_multiscan_mapper* mapper, const void* raw_xyz, uint32_t count,
                              uint32_t stride, loamx_cloud* full, uint32_t* ring_size, loamx_cloud* sharp, loamx_cloud* less_sharp,
                              loamx_cloud* flat, loamx_cloud* less_flat);

/* IMU data for the scan registration: updateIMUData (BasicScanRegistration.cpp:82-98) feeds the
 * handle's IMU history (capacity max(200, imu_history_size): the reference's buffer is created with 200 entries and
 * ensureCapacity only grows it, include/loam_velodyne/CircularBuffer.h); loamx_scanreg_set_time gives the scanTime of the
 * next process call (times in seconds on one clock); with a non-empty history loamx_scanreg_process_raw de-skews every
 * kept point (projectPointToStartOfSweep :101-147) and loamx_scanreg_get_imu_trans returns imuTransform() (:258-281):
 * start angles, current angles, position shift, velocity change — what loamx_odom_update_imu consumes.
 * acc_xyz: local acceleration with gravity removed and axes remapped as ScanRegistration.cpp:171-174 does. */
int loamx_scanreg_update_imu(loamx_scanreg* h, double stamp_sec, float roll, float pitch, float yaw, const float acc_xyz[3]);
int loamx_scanreg_set_time(loamx_scanreg* h, double scan_time_sec);
int loamx_scanreg_get_imu_trans(loamx_scanreg* h, float imu_trans[12]);

/* ------------------------------------------------------------------------------------------------------------
 * Sweep-to-sweep odometry  (BasicLaserOdometry)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_odom loamx_odom;

typedef struct loamx_odom_config {
  float scan_period;  /* 0.1  */
  int max_iterations; /* 25; 1 .. 255 */
  float delta_t_abort; /* 0.1 */
  float delta_r_abort; /* 0.1 */
  int device;
} loamx_odom_config;

void loamx_odom_default_config(loamx_odom_config* cfg);
loamx_odom* loamx_odom_create(const loamx_odom_config* cfg);
void loamx_odom_destroy(loamx_odom* h);
/* updateIMU: 4 x (x,y,z) = pitch/yaw/roll start, pitch/yaw/roll end, shift from start, velocity from start */
int loamx_odom_update_imu(loamx_odom* h, const float imu_trans[12]);
/* process(): the four feature clouds of the current sweep.  The first call only initialises (reference
 * BasicLaserOdometry.cpp:198-211). */
int loamx_odom_process(loamx_odom* h, const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat,
                       const loamx_cloud* less_flat);
int loamx_odom_get_transform(loamx_odom* h, float transform[6]);
int loamx_odom_get_transform_sum(loamx_odom* h, float transform_sum[6]);
int loamx_odom_set_transform(loamx_odom* h, const float transform[6]);
int loamx_odom_set_transform_sum(loamx_odom* h, const float transform_sum[6]);
/* lastCornerCloud()/lastSurfaceCloud(): the re-projected feature clouds handed on to mapping */
int loamx_odom_get_last_clouds(loamx_odom* h, loamx_cloud* last_corner, loamx_cloud* last_surf);
/* transformToEnd(cloud): in place on a caller cloud, with the current transform */
int loamx_odom_transform_to_end(loamx_odom* h, loamx_cloud* cloud);
/* diagnostics: iterations entered / rows selected in the last iteration of the last process() */
int loamx_odom_get_stats(loamx_odom* h, int stats[4]);

/* ------------------------------------------------------------------------------------------------------------
 * Linked nodes (ABI version 6): the reference's three nodes exchange a sweep's clouds as ROS messages — host memory
 * (ScanRegistration.cpp:186-208 -> LaserOdometry.cpp:141-233 -> LaserMapping.cpp:155-230).  A process that hosts the three
 * handles on ONE device can hand the clouds from node to node in HBM instead: same data flow, same results bit for bit
 * (tests/test_gpu_linked.py), without the five round trips over PCIe per sweep.
 *   loamx_scanreg_process_linked   the sweep goes up and the extraction is enqueued; returns without waiting.  The feature
 *                                  clouds and the sweep stay in the handle's device buffers.  A sweep of packed {stride 16,
 *                                  intensity at 12} records in memory the HIP runtime has pinned (hipHostMalloc / hipHostRegister)
 *                                  is copied by DMA from where it lies and must stay unchanged until loamx_odom_process_linked
 *                                  has returned; any other cloud has been copied out when this call returns.
 *   loamx_odom_process_linked      waits for `sr`'s extraction (and reports what loamx_scanreg_process would: LOAMX_E_INVALID
 *                                  for non-finite input, ...) — in two steps: the iterations start on the sharp / less-sharp / flat
 *                                  clouds, the less-flat cloud (its voxel grid still running) is taken when the sweep's tail
 *                                  needs it — runs process() on its clouds and re-projects the sweep's
 *                                  full-resolution cloud to the sweep end (transformToEnd of LaserOdometry.cpp:326) into a
 *                                  device buffer of `od`.  Returns with the pose (LOAMX_SKIPPED for the initialising sweep); the
 *                                  tail goes on behind it.  `sr`'s buffers are read until loamx_odom_link_wait(od) or
 *                                  loamx_map_process_linked has returned: start `sr`'s next sweep only then.
 *   loamx_map_process_linked       updateOdometry(od's transformSum) + process() on od's last corner / surface clouds and the
 *                                  re-projected full-resolution cloud, ordered behind od's tail on the device.
 *                                  full_res_registered (may be NULL): receives the registered full-resolution cloud
 *                                  (count = capacity in, points out; packed records in pinned memory receive it by DMA directly —
 *                                  loamx_map_process's full_res likewise).  `od` may start its next sweep when this has returned.
 * The host-cloud getters (loamx_odom_get_last_clouds, ...) keep working after the linked calls. */
int loamx_scanreg_process_linked(loamx_scanreg* h, const loamx_cloud* cloud, const uint32_t* ring_size, uint32_t n_rings);
int loamx_odom_process_linked(loamx_odom* h, loamx_scanreg* sr);
int loamx_odom_link_wait(loamx_odom* h);

/* ------------------------------------------------------------------------------------------------------------
 * Scan-to-map registration  (BasicLaserMapping)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_map loamx_map;

typedef struct loamx_map_config {
  float scan_period;     /* 0.1  */
  int max_iterations;    /* 10   */
  float delta_t_abort;   /* 0.05 */
  float delta_r_abort;   /* 0.05 */
  float corner_filter_size; /* 0.2 */
  float surf_filter_size;   /* 0.4 */
  float map_filter_size;    /* kept for API parity; unused by the reference as well (BasicLaserMapping.cpp:261) */
  int device;
} loamx_map_config;

void loamx_map_default_config(loamx_map_config* cfg);
loamx_map* loamx_map_create(const loamx_map_config* cfg);
void loamx_map_destroy(loamx_map* h);
int loamx_map_update_odometry(loamx_map* h, const float transform_sum[6]);
/* process(): corner_last / surf_last in, full_res registered in place (transformFullResToMap).
 * Returns with the sweep's results — the transforms, the statistics, the registered cloud.  The map update that ends the reference's
 * process() (insertion into the cubes and their re-filtering, BasicLaserMapping.cpp:535-593) is enqueued behind the registration on
 * the device and may still be running then: every later call that reads or changes the map (the next process / insert, get_cubes,
 * save_snapshot, ...) waits for it first, so the map any call sees is the reference's.  A failure inside the deferred update is
 * reported by that later call.  Every fifth processed frame (the surround cloud is cut from the updated map, :242-264) waits itself. */
int loamx_map_process(loamx_map* h, const loamx_cloud* corner_last, const loamx_cloud* surf_last, loamx_cloud* full_res);
/* see "Linked nodes" above */
int loamx_map_process_linked(loamx_map* h, loamx_odom* od, loamx_cloud* full_res_registered);
/* Behind a sweep's map update the handle prepares the NEXT sweep's map partition and sub-map index for the pose it predicts (constant
 * odometry velocity); the next process() adopts that work when the plan of the true pose — cube window, valid cubes, their order and
 * sizes — equals the predicted one entry by entry, and partitions afresh otherwise: results never depend on it.  counts[0] = sweeps
 * that adopted a prepared partition, counts[1] = sweeps whose prediction missed.  LOAMX_MAP_NO_SPECULATION=1 switches it off. */
int loamx_map_get_speculation(loamx_map* h, uint64_t counts[2]);
/* The map side of process() alone — the merge step of a map epoch (collective 3): a sweep that was registered elsewhere
 * (the batched pipeline, against a frozen copy of this map) is stacked, down-sized, inserted into the cubes with the GIVEN pose
 * (rx, ry, rz, tx, ty, tz = its transformAftMapped) and the touched cubes are re-filtered, exactly as process() does after its
 * optimisation (BasicLaserMapping.cpp:512-593); no optimisation runs, and the pose is inserted AS GIVEN — no IMU blend is applied to it even
 * when the handle holds IMU history (a pipeline pose has had its blend), and the handle's own transforms (Tobe / Bef / AftMapped), frame
 * counter and iteration limit are as before the call, also after an error.  loamx_map_get_cubes() afterwards is the next epoch's map
 * (loamx_pipeline_stage_frozen_* / loamx_dist_broadcast_map). */
int loamx_map_insert(loamx_map* h, const loamx_cloud* corner_last, const loamx_cloud* surf_last, const float pose6[6]);
/* which: 0 transformAftMapped, 1 transformBefMapped, 2 transformTobeMapped, 3 transformSum */
int loamx_map_get_transform(loamx_map* h, int which, float transform[6]);
int loamx_map_set_transform(loamx_map* h, int which, const float transform[6]);
/* updateIMU(IMUState2) (BasicLaserMapping.cpp:602-605, .h:47-75) and the laserOdometryTime argument of process()
 * (:266): with a non-empty IMU history transformUpdate blends 0.2 % of the interpolated IMU roll / pitch into the pose
 * (:173-200).  Times in seconds on a common clock. */
int loamx_map_update_imu(loamx_map* h, double stamp_sec, float roll, float pitch);
int loamx_map_set_time(loamx_map* h, double laser_odometry_time_sec);
int loamx_map_has_fresh_map(loamx_map* h);
/* laserCloudSurroundDS() */
int loamx_map_get_surround(loamx_map* h, loamx_cloud* out);
/* test / warm-start hook: insert map-frame points straight into the rolling cube grid (by coordinate) */
int loamx_map_load_cubes(loamx_map* h, const loamx_cloud* corner, const loamx_cloud* surf);
/* dump the whole map: which 0 = corner cubes, 1 = surf cubes */
int loamx_map_get_cubes(loamx_map* h, int which, loamx_cloud* out);
/* Map snapshot on disk (checkpointing a map, e.g. as the frozen map of the batched mode): the rolling map
 * of both feature types in storage order, the cube window, the frame counters and the five transforms.  A handle restored with
 * load continues bit for bit like the one that saved (same map filter sizes required).  File layout: mapping.hip, SnapshotHeader. */
int loamx_map_save_snapshot(loamx_map* h, const char* path);
int loamx_map_load_snapshot(loamx_map* h, const char* path);
/* diagnostics of the last process(): iterations, rows selected, corner queries, surf queries, corner sub-map size,
 * surf sub-map size, degenerate flag, optimised flag */
/* HIP-event timing of the registration inside the last process() (as loamx_batch_set_timing / loamx_batch_get_timing: ms[0] = the
 * registration's device time, ms[1] = sum of the Gauss-Newton launches, counts[0] = launches, counts[1] = query-iterations) */
int loamx_map_set_timing(loamx_map* h, int on);
int loamx_map_get_timing(loamx_map* h, float ms[4], uint64_t counts[4]);
int loamx_map_get_stats(loamx_map* h, int stats[8]);

/* ------------------------------------------------------------------------------------------------------------
 * Pose fusion after the path  (BasicTransformMaintenance, include/loam_velodyne/BasicTransformMaintenance.h:44-66,
 * src/lib/BasicTransformMaintenance.cpp:46-178) and the orientation convention of the nodes' nav_msgs/Odometry
 * messages (src/lib/LaserOdometry.cpp:300-308, src/lib/LaserMapping.cpp:205-213, src/lib/TransformMaintenance.cpp:
 * 66-115).  Host arithmetic only: these entry points need no device.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_tm loamx_tm;
loamx_tm* loamx_tm_create(void);
void loamx_tm_destroy(loamx_tm* h);
int loamx_tm_update_odometry(loamx_tm* h, const float transform_sum[6]);                                  /* updateOdometry */
int loamx_tm_update_mapping_transform(loamx_tm* h, const float aft_mapped[6], const float bef_mapped[6]); /* updateMappingTransform */
int loamx_tm_associate_to_map(loamx_tm* h);                                                                /* transformAssociateToMap */
int loamx_tm_get_mapped(loamx_tm* h, float transform_mapped[6]);                                           /* transformMapped() */
/* (rot_x, rot_y, rot_z) -> message orientation (x, y, z, w), and back */
int loamx_wire_pose_to_quat(const float rot_xyz[3], double quat_xyzw[4]);
int loamx_wire_quat_to_pose(const double quat_xyzw[4], float rot_xyz[3]);

/* ------------------------------------------------------------------------------------------------------------
 * Batched-sweep mode: B independent sweeps registered against one frozen sub-map.
 *   set_frozen[_device]  -> upload (or adopt) the sub-map and build its spatial index once per map epoch
 *   upload               -> stage the B sweeps' corner_last / surf_last (+ optional full-resolution) clouds in HBM
 *   run                  -> device only: stack round trip, voxel down-sampling, <= max_iterations Gauss-Newton
 *                           iterations per sweep, registration of the full-resolution clouds
 *   download             -> poses + per-sweep stats (+ optionally the registered clouds)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_batch loamx_batch;

loamx_batch* loamx_batch_create(const loamx_map_config* cfg, uint32_t max_sweeps);
void loamx_batch_destroy(loamx_batch* h);
int loamx_batch_set_frozen(loamx_batch* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map);
/* device-resident packed float4 (x,y,z,intensity) arrays, e.g. the buffers an RCCL broadcast just filled */
int loamx_batch_set_frozen_device(loamx_batch* h, const void* d_corner_xyzi, uint32_t n_corner, const void* d_surf_xyzi,
                                  uint32_t n_surf);
/* sweep s uses corner_last[s], surf_last[s], full_res[s] (full_res may be NULL) and guess[s][6] =
 * initial transformTobeMapped */
/* Double-buffered map epochs (BASELINE configs[4]): index the NEXT epoch's sub-map on a stream of its own
 * while sweeps are still registered against the current one, then make it current at a batch boundary.  The device
 * buffers are only read until the staged build has finished (hipDeviceSynchronize / the next swap), the index keeps a
 * cell-sorted copy.  wait_event (may be NULL): a hipEvent_t recorded behind whatever fills the buffers (a copy, an RCCL
 * broadcast) — the build is ordered behind it on the device, the host does not wait.  swap returns LOAMX_SKIPPED when nothing is staged. */
int loamx_batch_stage_frozen_device(loamx_batch* h, const void* d_corner_xyzi, uint32_t n_corner, const void* d_surf_xyzi,
                                    uint32_t n_surf, void* wait_event /* hipEvent_t of the buffers' producer, or NULL */);
int loamx_batch_stage_frozen(loamx_batch* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map);   /* host clouds */
int loamx_batch_swap_frozen(loamx_batch* h);
int loamx_batch_upload(loamx_batch* h, uint32_t n_sweeps, const loamx_cloud* corner_last, const loamx_cloud* surf_last,
                       const loamx_cloud* full_res, const float* guess6);
int loamx_batch_run(loamx_batch* h);
int loamx_batch_run_async(loamx_batch* h);
int loamx_batch_sync(loamx_batch* h);
/* poses6: n_sweeps x 6; stats: n_sweeps x 4 ints (iterations, rows selected, corner queries, surf queries);
 * either may be NULL */
int loamx_batch_download(loamx_batch* h, float* poses6, int* stats4);
int loamx_batch_download_full_res(loamx_batch* h, uint32_t sweep, loamx_cloud* out);
/* kernel-level timing of the last run(): ms[0] = whole run, ms[1] = sum of residual-kernel launches,
 * counts[0] = residual launches, counts[1] = total query-iterations executed, counts[2] = total DS queries */
/* record HIP events around every residual launch of the next run()s (off by default) */
int loamx_batch_set_timing(loamx_batch* h, int on);
int loamx_batch_get_timing(loamx_batch* h, float ms[4], uint64_t counts[4]);
/* raw HIP stream of the handle (hipStream_t) so a harness can bracket it with its own events */
void* loamx_batch_stream(loamx_batch* h);
/* Parity hook for the kNN contract (nanoflann_pcl.h:140-152, nanoflann.hpp:115-139, :372-379): runs the library's own
 * neighbour search (the device routine the Gauss-Newton kernel calls) for n query points given in the MAP frame against
 * the frozen corner (which = 0) or surf (which = 1) sub-map.  idx5[5 q + k] = index, in the cloud handed to set_frozen, of
 * the k-th nearest point of query q, d2_5 = its squared distance in float, (dx*dx + dy*dy) + dz*dz; ascending by
 * (distance, index).  Only neighbours closer than 1.05 m are searched for (the reference rejects a query whose fifth
 * neighbour is 1 m away or more, BasicLaserMapping.cpp:671, :760): missing entries are 0xffffffff / FLT_MAX. */
int loamx_batch_knn_probe(loamx_batch* h, int which, const float* queries_xyz, uint32_t n, uint32_t* idx5, float* d2_5);
/* Parity hook for the 6x6 solve of the update steps (colPivHouseholderQr().solve: BasicLaserMapping.cpp:867, BasicLaserOdometry.cpp:559):
 * n systems ata[36 i .. ) x = atb[6 i .. ) through the wave-cooperative routine the kernels call (x_coop) and through the scalar
 * routine it must equal bit for bit (x_scalar, one thread, the reference's order of operations). */
int loamx_batch_qr6_probe(loamx_batch* h, const float* ata, const float* atb, uint32_t n, float* x_coop, float* x_scalar);
/* Stress probe of the exchange between the workgroups of one odometry stream (tagged 16-byte records written by one agent-scope store,
 * accepted by the reader when both tags match): `pairs` producer / consumer workgroup pairs on different XCDs, `rounds` versions per
 * record.  out4 = {accepted reads, torn reads (tags differ: the reader polls again), INCONSISTENT accepted reads (must be 0: the
 * property the exchange rests on), consumer threads that gave up waiting (must be 0)}. */
int loamx_batch_xrec_stress(loamx_batch* h, uint32_t pairs, uint32_t rounds, uint64_t out4[4]);
/* Parity hook for the voxel-grid stage: the down-sampled stack clouds of one sweep of the last run (laserCloudCornerStackDS /
 * laserCloudSurfStackDS, BasicLaserMapping.cpp:512-527 — the query points of the Gauss-Newton iterations, sensor frame, in
 * pcl::VoxelGrid's output order).  count fields: capacity in, size out; LOAMX_E_CAPACITY when a cloud does not fit. */
int loamx_batch_download_ds(loamx_batch* h, uint32_t sweep, loamx_cloud* corner_ds, loamx_cloud* surf_ds);

/* ------------------------------------------------------------------------------------------------------------
 * Streaming pipeline: n independent streams, each advancing one sweep per step through feature extraction ->
 * odometry -> registration against the frozen sub-map.  A stream keeps the reference's sequential state (odometry
 * transform / transformSum / last clouds, mapping transformBefMapped / transformAftMapped); only the map is frozen.
 * Sweeps are staged in HBM ahead of time, so step() is device work plus a few offset / pose read-backs.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_pipeline loamx_pipeline;

loamx_pipeline* loamx_pipeline_create(const loamx_scanreg_config* fcfg, const loamx_odom_config* ocfg,
                                      const loamx_map_config* mcfg, uint32_t n_streams);
void loamx_pipeline_destroy(loamx_pipeline* h);
int loamx_pipeline_set_frozen(loamx_pipeline* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map);
int loamx_pipeline_set_frozen_device(loamx_pipeline* h, const void* d_corner_xyzi, uint32_t n_corner, const void* d_surf_xyzi,
                                     uint32_t n_surf);
/* double-buffered map epochs, as loamx_batch_stage_frozen_device / loamx_batch_swap_frozen; swap between two steps */
int loamx_pipeline_stage_frozen_device(loamx_pipeline* h, const void* d_corner_xyzi, uint32_t n_corner, const void* d_surf_xyzi,
                                       uint32_t n_surf, void* wait_event);
int loamx_pipeline_stage_frozen(loamx_pipeline* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map);
int loamx_pipeline_swap_frozen(loamx_pipeline* h);
/* seed a stream's state; any pointer may be NULL (left unchanged) */
int loamx_pipeline_set_state(loamx_pipeline* h, uint32_t stream, const float* transform, const float* transform_sum,
                             const float* bef_mapped, const float* aft_mapped);
/* stage n_steps sweeps per stream: sweep (t, s) = clouds[t * n_streams + s] (rings concatenated) */
int loamx_pipeline_upload(loamx_pipeline* h, uint32_t n_steps, const loamx_cloud* clouds, const uint32_t* const* ring_size,
                          const uint32_t* n_rings);
/* Streaming input (the PCIe-inclusive mode, SURVEY.md §8d "GPU timing"): instead of staging a whole run with
 * loamx_pipeline_upload, hand over ONE step at a time, in order, without blocking — the copies go to a stream of their own
 * (packed float4 clouds {stride 16, intensity at 12} straight from the caller's memory: pin it, e.g. hipHostRegister, and the
 * transfer is a DMA that overlaps the kernels of the steps in flight; other layouts are repacked through pinned staging on
 * the calling thread).  Up to eight steps are in flight: stage_step(t) may be called once step(t - 8) has returned; the
 * buffers of step t are read until step(t) has returned. */
int loamx_pipeline_stage_step(loamx_pipeline* h, uint32_t step, const loamx_cloud* clouds, const uint32_t* const* ring_size,
                              const uint32_t* n_rings);
/* Raw input for the batched pipeline: step `step` as the sensors delivered it — raw_xyz[s] = one
 * revolution of stream s, counts[s] records with x, y, z float32 at byte offsets 0 / 4 / 8, `stride` bytes apart, sensor axes,
 * firing order (the /velodyne_points payload; MultiScanRegistration.cpp:160-238).  The payloads cross PCIe as they are (pin them
 * for a DMA) and are re-strided, binned into rings and — for a stream with IMU data — de-skewed on the device; same ordering
 * rules as loamx_pipeline_stage_step.  scan_time_sec[s] (may be NULL without IMU data) is the sweep's time stamp on the clock of
 * loamx_pipeline_update_imu, which feeds stream s's IMU history exactly as loamx_scanreg_update_imu does (updateIMUData,
 * BasicScanRegistration.cpp:82-98); the resulting imuTransform() of every sweep is plugged into that stream's odometry
 * (BasicLaserOdometry::updateIMU).  The same messages feed the stream's mapping-side history (LaserMapping's own subscription, 200
 * deep): a sweep staged with a time stamp gets transformUpdate's roll / pitch blend (BasicLaserMapping.cpp:171-200) from the messages
 * that had arrived when it was staged, before its full-resolution cloud is registered. */
int loamx_pipeline_stage_step_raw(loamx_pipeline* h, uint32_t step, const void* const* raw_xyz, const uint32_t* counts, uint32_t stride,
                                  const loamx_multiscan_mapper* mapper, const double* scan_time_sec);
int loamx_pipeline_update_imu(loamx_pipeline* h, uint32_t stream, double stamp_sec, float roll, float pitch, float yaw, const float acc_xyz[3]);
/* Asynchronous output: after enable (before the first step), download_step_async — called after loamx_pipeline_step(t) —
 * starts copying the registered full-resolution clouds of step t (out[k] = k-th stream that was registered; packed float4
 * records; count in = capacity, out = points) on a copy stream and returns; the registration alternates between two device
 * buffers so the next step does not wait for the copy.  wait_downloads blocks until every started copy has landed.
 * When the destination is pinned memory of the ROCm runtime's own (hipHostMalloc, torch's pinned tensors) the copy is handed to
 * the GPU's SDMA engine directly; other host memory goes through hipMemcpyAsync on a copy stream (whose choice of engine may
 * disturb kernels that write to host memory, see csrc/hostlink.hpp).  download_counts: [0] downloads issued the first way,
 * [1] the second. */
int loamx_pipeline_enable_async_downloads(loamx_pipeline* h);
int loamx_pipeline_download_step_async(loamx_pipeline* h, loamx_cloud* out, uint32_t n_out);
int loamx_pipeline_wait_downloads(loamx_pipeline* h);
int loamx_pipeline_download_counts(loamx_pipeline* h, uint64_t counts[2]);
/* run staged step t for every stream.  LOAMX_SKIPPED when no stream reached the registration stage (first sweeps) */
int loamx_pipeline_step(loamx_pipeline* h, uint32_t step);
/* stats8: odometry iterations, odometry rows, mapping iterations, mapping rows, corner queries, surf queries,
 * degenerate, mapped */
int loamx_pipeline_get(loamx_pipeline* h, uint32_t stream, float* transform, float* transform_sum, float* aft_mapped,
                       int* stats8);
/* registered full-resolution cloud of the k-th stream that was registered in the last step */
int loamx_pipeline_download_full_res(loamx_pipeline* h, uint32_t slot, loamx_cloud* out);
/* The re-projected less-sharp / less-flat clouds of one stream's sweep of the LAST step (laserCloudCornerLast / laserCloudSurfLast as the
 * odometry hands them to the mapping, LaserOdometry.cpp:296-326): with the stream's transformAftMapped (loamx_pipeline_get) they are what
 * loamx_map_insert needs to merge the sweep into the next epoch's map.  count fields: capacity in, size out. */
int loamx_pipeline_download_last_clouds(loamx_pipeline* h, uint32_t stream, loamx_cloud* last_corner, loamx_cloud* last_surf);
/* Look-ahead (default on): while step t's registration runs, the odometry and the feature extraction of the following staged
 * steps already execute on their own HIP streams (they are independent ROS nodes in the reference) — up to
 * loamx_pipeline_lookahead_depth() steps ahead.  Results are identical; loamx_pipeline_get() always reports the sweep that was
 * registered last.  Turn it off when per-stream state is changed with loamx_pipeline_set_state() between steps. */
int loamx_pipeline_set_lookahead(loamx_pipeline* h, int on);
/* How many steps beyond the one being registered the odometry may run: 0 with the look-ahead off, 6 for batches staged with
 * loamx_pipeline_upload (a stream whose sweeps need every odometry iteration needs them for several sweeps in a row; the lead absorbs
 * such a run) and for the streaming ring (loamx_pipeline_stage_step*: eight slots, i.e. six steps beyond the one being registered
 * and one more being staged). */
int loamx_pipeline_lookahead_depth(loamx_pipeline* h);
/* Blocks until the look-ahead has finished every step it is currently allowed to run ahead (odometry of up to
 * loamx_pipeline_lookahead_depth() steps beyond the last loamx_pipeline_step, their feature extraction) and everything of it is enqueued on the device; *last_odometry_step (may be NULL)
 * receives the last step whose odometry is complete.  For a caller that wants a quiescent pipeline — a benchmark window that must
 * contain the look-ahead work it profits from, a clean shutdown point — without switching the look-ahead off. */
int loamx_pipeline_drain_lookahead(loamx_pipeline* h, int* last_odometry_step);
/* HIP-event timing of the stages: 0 off, 1 stage events + an event pair around every Gauss-Newton launch, 2 stage events only
 * (the pairs cost ~3 % of a step: a caller that wants both the rate and the launch durations samples them, as bench.py does). */
int loamx_pipeline_set_timing(loamx_pipeline* h, int on);
/* stage_ms: features, odometry, registration, whole step (HIP events on the pipeline's stream);
 * reg_ms / counts as loamx_batch_get_timing */
int loamx_pipeline_get_timing(loamx_pipeline* h, float stage_ms[4], float reg_ms[4], uint64_t counts[4]);
/* The odometry chains' launch pairs (k_odom_corr_grid + k_odom_lm, BasicLaserOdometry.cpp:246-622 in groups of five iterations), timed
 * with HIP events on the chains' own streams while loamx_pipeline_set_timing(h, 1) is on; running totals over all chains since the handle
 * was created (never waits: a call whose events have not completed is counted later).
 * ms4 = {k_odom_lm launches that iterated, k_odom_lm launches over converged streams, k_odom_corr_grid likewise x 2};
 * counts7 = {lm launches that iterated, lm no-op launches, iterations (slowest stream of each launch, summed), corr launches that
 * searched, corr no-op launches, algorithmic bytes of the iterating lm launches (48 B x features of the streams still iterating),
 * features searched by the corr launches}. */
int loamx_pipeline_get_odom_launch_timing(loamx_pipeline* h, double ms4[4], uint64_t counts7[7]);
void* loamx_pipeline_stream(loamx_pipeline* h);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, RCCL over xGMI.  The batched mode shards by independent sweeps — rank r of G
 * registers sweeps [r*B/G, (r+1)*B/G) against a replica of the frozen map — so the data path has no collective.  The two
 * exchanges are (1) the map epoch: ncclBroadcast of the two sub-map buffers, asynchronous, returning an event that
 * loamx_{batch,pipeline}_stage_frozen_device takes as wait_event (the index build of epoch k+1 is ordered behind the broadcast
 * on the device while epoch k's registrations run: double buffering), and (2) the results: ncclAllGather of
 * n_local x (6 pose floats + iterations + flags) per rank.  The 128-byte id from loamx_dist_get_unique_id (rank 0) reaches the
 * other processes by the host's own means (loam_velodyne_amd/launch.py uses a file).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct loamx_dist loamx_dist;
#define LOAMX_DIST_ID_BYTES 128
int loamx_dist_get_unique_id(unsigned char id[LOAMX_DIST_ID_BYTES]);
loamx_dist* loamx_dist_create(const unsigned char id[LOAMX_DIST_ID_BYTES], int rank, int world_size, int device);
void loamx_dist_destroy(loamx_dist* h);
int loamx_dist_rank(const loamx_dist* h);
int loamx_dist_world_size(const loamx_dist* h);
/* this rank's contiguous share [begin, end) of a batch of `batch` sweeps */
int loamx_dist_shard(const loamx_dist* h, uint32_t batch, uint32_t* begin, uint32_t* end);
/* device buffers of packed float4 (x,y,z,intensity), the same sizes on every rank; in place (root's content reaches all).
 * wait_event (may be NULL): hipEvent_t behind whatever fills the root's buffers; *done_event (may be NULL) receives a hipEvent_t
 * owned by the handle, recorded behind the broadcast — valid until the next broadcast on this handle. */
int loamx_dist_broadcast_map(loamx_dist* h, void* d_corner_xyzi, uint32_t n_corner, void* d_surf_xyzi, uint32_t n_surf, int root,
                             void* wait_event, void** done_event);
/* every rank contributes ITS n_local records (poses6[n_local][6], iters_flags2[n_local][2], the latter may be NULL; n_local may
 * differ between ranks and may be 0 — the shards of loamx_dist_shard are unequal whenever the batch does not divide by the world
 * size) and receives the records of all ranks concatenated in rank order (sum of the counts = the batch); counts_all (may be NULL)
 * receives the world_size record counts.  Every rank must call it; blocking. */
int loamx_dist_allgather_results(loamx_dist* h, const float* poses6, const int* iters_flags2, uint32_t n_local, float* poses6_all,
                                 int* iters_flags2_all, uint32_t* counts_all);
/* the same with the capacity of the receive arrays stated (in records): LOAMX_E_CAPACITY — after both collectives, so that no rank
 * is left waiting, and with counts_all filled — when the ranks' records do not fit; nothing is written beyond the capacity */
int loamx_dist_allgather_results_cap(loamx_dist* h, const float* poses6, const int* iters_flags2, uint32_t n_local, float* poses6_all,
                                     int* iters_flags2_all, uint32_t capacity_records, uint32_t* counts_all);
/* only the first half: every rank's record count (for a caller that sizes its receive arrays by them).  Every rank must call it. */
int loamx_dist_allgather_counts(loamx_dist* h, uint32_t n_local, uint32_t* counts_all);
/* ranks the RCCL communicator itself reports (ncclCommCount); -1 on error */
int loamx_dist_comm_count(loamx_dist* h);
/* The host side of the two rules above without a device or a communicator (for a host that brings its own transport; the
 * library's own all-gather uses exactly these): the shard of a rank; the padded send block of a rank (n_pad records of
 * LOAMX_DIST_RECORD_FLOATS floats = 6 pose floats + iterations + flags as raw ints, zero beyond n_local); and the inverse for the
 * gathered blocks recv8[world_size][n_pad][...] given every rank's record count. */
#define LOAMX_DIST_RECORD_FLOATS 8
int loamx_dist_shard_of(int rank, int world_size, uint32_t batch, uint32_t* begin, uint32_t* end);
int loamx_dist_pack_results(const float* poses6, const int* iters_flags2, uint32_t n_local, uint32_t n_pad, float* send8);
int loamx_dist_unpack_results(const float* recv8, const uint32_t* counts, int world_size, uint32_t n_pad, float* poses6_all,
                              int* iters_flags2_all);
/* The merge step of a map epoch (BasicLaserMapping.cpp:536-593 is what loamx_map_insert restates): the sweeps a rank has registered —
 * per stream the re-projected corner / surf clouds of its last step (loamx_pipeline_download_last_clouds) and its transformAftMapped —
 * travel to the rank that owns the map accumulator as ONE message of 32-bit words per rank:
 *   'LXCL', n_streams, (n_corner, n_surf) x n_streams | 6 pose floats x n_streams | the points (x y z intensity), corner(0) surf(0) ...
 * pack / unpack are host-side layout rules (no device, any transport); words == NULL asks for the size only.
 * loamx_dist_gatherv: variable-size gather to `root` over RCCL — every rank's word count first (all-gather), then the words point to
 * point inside one group; recv_words (root only) receives the ranks' messages back to back in rank order, counts_all (may be NULL)
 * every rank's word count.  Every rank calls it, also with n_words = 0. */
int loamx_dist_pack_clouds(uint32_t n_streams, const loamx_cloud* corner, const loamx_cloud* surf, const float* poses6, uint32_t* words,
                           uint64_t capacity_words, uint64_t* n_words);
int loamx_dist_unpack_clouds_header(const uint32_t* words, uint64_t n_words, uint32_t* n_streams, uint32_t* n_corner, uint32_t* n_surf,
                                    uint32_t capacity_streams);
int loamx_dist_unpack_clouds_stream(const uint32_t* words, uint64_t n_words, uint32_t stream, float pose6[6], loamx_cloud* corner,
                                    loamx_cloud* surf);
int loamx_dist_gatherv(loamx_dist* h, const uint32_t* send_words, uint32_t n_words, int root, uint32_t* recv_words, uint64_t capacity_words,
                       uint32_t* counts_all);
int loamx_dist_barrier(loamx_dist* h);
void* loamx_dist_stream(loamx_dist* h);   /* hipStream_t of the collectives */

#ifdef __cplusplus
}
#endif
#endif /* LOAMX_H */
