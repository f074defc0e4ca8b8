"""Named tuples of the data layer (API of /root/reference/src/data/basetypes.py:34-37)."""
from collections import namedtuple

DepthFrame = namedtuple('DepthFrame', ['dpt', 'gtorig', 'gtcrop', 'T', 'gt3Dorig', 'gt3Dcrop', 'com', 'fileName', 'subSeqName', 'side', 'extraData'])
NamedImgSequence = namedtuple('NamedImgSequence', ['name', 'data', 'config'])
