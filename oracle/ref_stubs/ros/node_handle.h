// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: see ros/ros.h in this directory.
#pragma once
#include <ros/ros.h>
