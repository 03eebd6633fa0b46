#pragma once
#include "hip_runtime.h"
