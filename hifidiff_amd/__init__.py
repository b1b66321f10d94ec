"""hifidiff_amd: MI355X-native refiner sampling path of HifiDiff."""
