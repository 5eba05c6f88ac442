"""MI355X-native Muskingum routing engine behind river-route's Router API."""
__version__ = '0.1.0'
