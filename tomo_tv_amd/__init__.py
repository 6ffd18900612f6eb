"""tomo_tv_amd -- MI355X-native hot path of jtschwar/tomo_TV behind the TomoGPU / tomoengine API."""
__version__ = "0.1.0"
