// api_ops.cpp -- placeholder, filled in below.
