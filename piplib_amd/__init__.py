"""piplib_amd -- MI355X-native PipLib hot path."""
