EMO_ABI_VERSION = 8   # must equal EMO_ABI_VERSION in include/emo_hip.h
