EMO_ABI_VERSION = 10   # must equal EMO_ABI_VERSION in include/emo_hip.h
