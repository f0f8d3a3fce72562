#pragma once
//! Drop-in include name kept from the reference; everything lives in map.hpp.
#include "map.hpp"
