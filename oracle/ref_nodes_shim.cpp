// ORACLE — TEST INFRASTRUCTURE ONLY.  The reference's FOUR NODES in one process: MultiScanRegistration -> LaserOdometry ->
// LaserMapping -> TransformMaintenance, the reference's own node classes (src/lib/*.cpp compiled where they lie) wired through
// the in-process topic bus of oracle/ref_stubs/ros/ros.h and driven by a deterministic schedule (every node keeps up with every
// message: deliver, process, deliver).  Built twice by the same recipe:
//   oracle/_ref/libref_nodes.so                       the wrappers + the reference's Basic*.cpp            (CPU, the checker)
//   oracle/_ref/libloam_nodes.so   the SAME wrappers + loamx_adapter.h + libloamx.so   (the product, needs a GPU)
// so a test can feed both the same /multi_scan_points and /imu/data messages and compare what comes out of /laser_odom_to_init,
// /aft_mapped_to_init and /integrated_to_init.  Stand-ins: see oracle/ref_stubs (ROS / tf / PCL / Eigen are absent from the image).
#include <algorithm>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>
#include "loam_velodyne/MultiScanRegistration.h"
#include "loam_velodyne/LaserOdometry.h"
#include "loam_velodyne/LaserMapping.h"
#include "loam_velodyne/TransformMaintenance.h"

namespace {

struct OdomRecord {
  double stamp;
  float v[13];   // orientation x y z w, position x y z, twist angular x y z, twist linear x y z
};
struct Collector {
  std::vector<OdomRecord> odom[3];                 // 0 /laser_odom_to_init, 1 /aft_mapped_to_init, 2 /integrated_to_init
  std::vector<std::vector<float>> clouds[2];       // 0 /velodyne_cloud_registered, 1 /laser_cloud_surround
  static OdomRecord rec(const nav_msgs::Odometry& m) {
    OdomRecord r;
    r.stamp = m.header.stamp.toSec();
    const auto& q = m.pose.pose.orientation;
    const auto& p = m.pose.pose.position;
    const auto& a = m.twist.twist.angular;
    const auto& l = m.twist.twist.linear;
    const double d[13] = {q.x, q.y, q.z, q.w, p.x, p.y, p.z, a.x, a.y, a.z, l.x, l.y, l.z};
    for (int k = 0; k < 13; k++) r.v[k] = (float)d[k];
    return r;
  }
  void onOdom(const nav_msgs::Odometry::ConstPtr& m) { odom[0].push_back(rec(*m)); }
  void onAft(const nav_msgs::Odometry::ConstPtr& m) { odom[1].push_back(rec(*m)); }
  void onIntegrated(const nav_msgs::Odometry::ConstPtr& m) { odom[2].push_back(rec(*m)); }
  void onRegistered(const sensor_msgs::PointCloud2ConstPtr& m) { clouds[0].push_back(m->data); }
  void onSurround(const sensor_msgs::PointCloud2ConstPtr& m) { clouds[1].push_back(m->data); }
};

struct Nodes {
  ros::NodeHandle node, priv;
  loam::MultiScanRegistration scan;
  loam::LaserOdometry odometry;
  loam::LaserMapping mapping;
  loam::TransformMaintenance maintenance;
  Collector out;
  ros::Publisher pubCloud, pubImu;
  bool ok = false;
  std::string error;          // what() of an exception a node threw (the product's adapter throws std::runtime_error on device errors)
  Nodes() : odometry(0.1f), mapping(0.1f) {}
};

}  // namespace

extern "C" {

// parameters of the (shared) private node handle, set BEFORE nodes_create: "lidar" = VLP-16 | HDL-32 | HDL-64E, "scanPeriod",
// "ioRatio", "maxIterations", "imuHistorySize", "cornerFilterSize", ... exactly the names the reference's setup() functions read
void nodes_reset_bus() { ros::bus::reset(); }
void nodes_set_param(const char* name, const char* value) { ros::bus::state().params[name] = value; }

void* nodes_create() {
  auto* n = new Nodes();
  try {
  n->ok = n->scan.setup(n->node, n->priv) && n->odometry.setup(n->node, n->priv) && n->mapping.setup(n->node, n->priv) &&
          n->maintenance.setup(n->node, n->priv);
  n->node.subscribe<nav_msgs::Odometry>("/laser_odom_to_init", 5, &Collector::onOdom, &n->out);
  n->node.subscribe<nav_msgs::Odometry>("/aft_mapped_to_init", 5, &Collector::onAft, &n->out);
  n->node.subscribe<nav_msgs::Odometry>("/integrated_to_init", 5, &Collector::onIntegrated, &n->out);
  n->node.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_registered", 2, &Collector::onRegistered, &n->out);
  n->node.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_surround", 1, &Collector::onSurround, &n->out);
  n->pubCloud = n->node.advertise<sensor_msgs::PointCloud2>("/multi_scan_points", 2);
  n->pubImu = n->node.advertise<sensor_msgs::Imu>("/imu/data", 50);
  // the scan registration swallows its first 20 cloud messages (MultiScanRegistration.cpp:143-149): get that over with
  for (int k = 0; k < 20; k++) {
    n->pubCloud.publish(sensor_msgs::PointCloud2());
    ros::spinOnce();
  }
  } catch (const std::exception& e) { n->error = e.what(); n->ok = false; }
  return n;
}
int nodes_ok(void* h) { return ((Nodes*)h)->ok ? 1 : 0; }
void nodes_destroy(void* h) { delete (Nodes*)h; }

// one /imu/data message: orientation quaternion (x, y, z, w), linear acceleration incl. gravity, in the IMU's axes
const char* nodes_last_error(void* h) { return ((Nodes*)h)->error.c_str(); }

int nodes_push_imu(void* h, unsigned sec, unsigned nsec, const double* q4, const double* acc3) {
  auto* n = (Nodes*)h;
  try {
  sensor_msgs::Imu m;
  m.header.stamp.sec = sec; m.header.stamp.nsec = nsec;
  m.orientation.x = q4[0]; m.orientation.y = q4[1]; m.orientation.z = q4[2]; m.orientation.w = q4[3];
  m.linear_acceleration.x = acc3[0]; m.linear_acceleration.y = acc3[1]; m.linear_acceleration.z = acc3[2];
  n->pubImu.publish(m);
  ros::spinOnce();      // ScanRegistration::handleIMUMessage and LaserMapping::imuHandler
  } catch (const std::exception& e) { n->error = e.what(); return -1; }
  return 0;
}

// one /multi_scan_points message (n x (x, y, z) in the sensor's axes, firing order) pushed through all four nodes
int nodes_push_cloud(void* h, const float* raw, int n_pts, unsigned sec, unsigned nsec) {
  auto* n = (Nodes*)h;
  try {
  sensor_msgs::PointCloud2 m;
  m.header.stamp.sec = sec; m.header.stamp.nsec = nsec;
  m.floats_per_point = 3;
  m.data.assign(raw, raw + 3 * (size_t)n_pts);
  n->pubCloud.publish(m);
  ros::spinOnce();            // scan registration runs in its callback; its six topics reach the odometry's handlers
  n->odometry.process();      // LaserOdometry::process (needs all six fresh); publishes odometry (+ clouds every ioRatio-th sweep)
  ros::spinOnce();            // -> mapping handlers, transform maintenance (which publishes /integrated_to_init)
  n->mapping.process();       // LaserMapping::process; publishes /aft_mapped_to_init, the registered cloud, the surround cloud
  ros::spinOnce();            // -> transform maintenance
  } catch (const std::exception& e) { n->error = e.what(); return -1; }
  return 0;
}

// which: 0 /laser_odom_to_init, 1 /aft_mapped_to_init, 2 /integrated_to_init; out: count x (stamp as double is returned separately)
int nodes_odom_count(void* h, int which) { return (int)((Nodes*)h)->out.odom[which].size(); }
void nodes_odom_get(void* h, int which, int i, double* stamp, float* v13) {
  const OdomRecord& r = ((Nodes*)h)->out.odom[which][i];
  *stamp = r.stamp;
  std::memcpy(v13, r.v, sizeof(r.v));
}
// which: 0 /velodyne_cloud_registered, 1 /laser_cloud_surround; returns floats in message i (4 per point); out may be NULL
int nodes_cloud_count(void* h, int which) { return (int)((Nodes*)h)->out.clouds[which].size(); }
int nodes_cloud_get(void* h, int which, int i, float* out, int cap_floats) {
  const std::vector<float>& c = ((Nodes*)h)->out.clouds[which][i];
  if (out) std::memcpy(out, c.data(), sizeof(float) * std::min((size_t)cap_floats, c.size()));
  return (int)c.size();
}

}  // extern "C"
