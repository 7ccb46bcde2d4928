// The ONE wrapper translation unit a maintainer swaps: loam::MultiScanRegistration and loam::MultiScanMapper as
// declared by the reference's include/loam_velodyne/MultiScanRegistration.h, with the per-point host loop of
// MultiScanRegistration::process (src/lib/MultiScanRegistration.cpp:160-238) replaced by one call into the device path —
// ring binning, relative times and the IMU de-skew run in loamx_scanreg_process_raw.  Parameter handling and the start-up delay
// follow the reference's setupROS / handleCloudMessage (:79-156) so that the node behaves the same on the ROS side.
#include "loam_velodyne/MultiScanRegistration.h"
#include <pcl_conversions/pcl_conversions.h>

namespace loam {

MultiScanMapper::MultiScanMapper(const float& lowerBound, const float& upperBound, const uint16_t& nScanRings)
    : _lowerBound(lowerBound), _upperBound(upperBound), _nScanRings(nScanRings), _factor((nScanRings - 1) / (upperBound - lowerBound)) {}

void MultiScanMapper::set(const float& lowerBound, const float& upperBound, const uint16_t& nScanRings) {
  *this = MultiScanMapper(lowerBound, upperBound, nScanRings);
}

MultiScanRegistration::MultiScanRegistration(const MultiScanMapper& scanMapper) : _scanMapper(scanMapper) {}

bool MultiScanRegistration::setup(ros::NodeHandle& node, ros::NodeHandle& privateNode) {
  RegistrationParams config;
  return setupROS(node, privateNode, config) && configure(config);
}

bool MultiScanRegistration::setupROS(ros::NodeHandle& node, ros::NodeHandle& privateNode, RegistrationParams& config_out) {
  if (!ScanRegistration::setupROS(node, privateNode, config_out)) return false;
  std::string lidar;
  if (privateNode.getParam("lidar", lidar)) {
    if (lidar == "VLP-16") _scanMapper = MultiScanMapper::Velodyne_VLP_16();
    else if (lidar == "HDL-32") _scanMapper = MultiScanMapper::Velodyne_HDL_32();
    else if (lidar == "HDL-64E") _scanMapper = MultiScanMapper::Velodyne_HDL_64E();
    else {
      ROS_ERROR("Invalid lidar parameter: %s (only \"VLP-16\", \"HDL-32\" and \"HDL-64E\" are supported)", lidar.c_str());
      return false;
    }
    if (!privateNode.hasParam("scanPeriod")) config_out.scanPeriod = 0.1;
  } else {
    float lo, hi;
    int rings;
    if (privateNode.getParam("minVerticalAngle", lo) && privateNode.getParam("maxVerticalAngle", hi) && privateNode.getParam("nScanRings", rings)) {
      if (lo >= hi || rings < 2) {
        ROS_ERROR("Invalid vertical range or number of scan rings");
        return false;
      }
      _scanMapper.set(lo, hi, rings);
    }
  }
  _subLaserCloud = node.subscribe<sensor_msgs::PointCloud2>("/multi_scan_points", 2, &MultiScanRegistration::handleCloudMessage, this);
  return true;
}

void MultiScanRegistration::handleCloudMessage(const sensor_msgs::PointCloud2ConstPtr& laserCloudMsg) {
  if (_systemDelay > 0) {   // the first 20 messages are dropped, as the reference does
    --_systemDelay;
    return;
  }
  pcl::PointCloud<pcl::PointXYZ> laserCloudIn;
  pcl::fromROSMsg(*laserCloudMsg, laserCloudIn);
  process(laserCloudIn, fromROSTime(laserCloudMsg->header.stamp));
}

void MultiScanRegistration::process(const pcl::PointCloud<pcl::PointXYZ>& laserCloudIn, const Time& scanTime) {
  processRawSweep(scanTime, laserCloudIn, _scanMapper.getLowerBound(), _scanMapper.getUpperBound(), _scanMapper.getNumberOfScanRings());
  publishResult();
}

}  // namespace loam
