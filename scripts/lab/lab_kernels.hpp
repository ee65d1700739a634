#pragma once
#include "mfma_kernels.hpp"
#include "lab_v3_gen.hpp"
#include <type_traits>
#include "lab_v5.hpp"
#include "lab_v6.hpp"
#include "lab_v7.hpp"
#include "lab_v8.hpp"
