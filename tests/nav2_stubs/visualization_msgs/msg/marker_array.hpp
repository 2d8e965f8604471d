#pragma once
#include <vector>
namespace visualization_msgs { namespace msg { struct Marker {}; struct MarkerArray { std::vector<Marker> markers; }; } }
