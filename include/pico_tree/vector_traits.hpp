#pragma once
//! Drop-in include name kept from the reference; everything lives in traits.hpp.
#include "traits.hpp"
