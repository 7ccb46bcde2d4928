// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: plain message structs.
#pragma once
namespace geometry_msgs {
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct Twist { Vector3 linear, angular; };
struct PoseWithCovariance { Pose pose; };
struct TwistWithCovariance { Twist twist; };
}  // namespace geometry_msgs
