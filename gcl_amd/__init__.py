"""gcl_amd: MI355X-native hot path of liuQuan98/GCL."""
