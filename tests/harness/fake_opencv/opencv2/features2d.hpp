#include "opencv2/core.hpp"  // fake, see opencv2/core.hpp
