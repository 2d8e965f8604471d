#pragma once
#include <string>
#include <vector>
#include "geometry_msgs/msg/pose_stamped.hpp"
namespace nav2_costmap_2d {
class Costmap2D {
 public:
  unsigned char *getCharMap() const { return nullptr; }
  unsigned int getSizeInCellsX() const { return 0; }
  unsigned int getSizeInCellsY() const { return 0; }
  double getOriginX() const { return 0; }
  double getOriginY() const { return 0; }
  double getResolution() const { return 0; }
};
class Costmap2DROS {
 public:
  Costmap2D *getCostmap() { return nullptr; }
  std::vector<geometry_msgs::msg::Point> getRobotFootprint() { return {}; }
  std::string getGlobalFrameID() { return ""; }
};
}  // namespace nav2_costmap_2d
