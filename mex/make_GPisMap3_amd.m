% make_GPisMap3_amd.m -- builds the reference's UNCHANGED 3-D gateway (mex/mexGPisMap3.cpp of
% leebhoram/GPisMap) against the MI355X-native library instead of the Eigen sources.
% Twin of the reference's mex/make_GPisMap3.m:1-15: same gateway file, same CXXFLAGS; the six
% reference .cpp files and the Eigen include path are replaced by include/ + libgpismap_amd.so.
%
% Usage (from the reference's mex/ directory, after building the library with
%   python -c "import __graft_entry__ as g; g.build()"   in the gpismap_amd checkout):
%   >> GPISMAP_AMD = '/path/to/gpismap_amd_checkout';  run([GPISMAP_AMD '/mex/make_GPisMap3_amd.m'])
% The resulting mexGPisMap3.<mexext> is what matlab/demo_gpisMap3.m and
% matlab/plot_scripts/visualize_gpisMap3.m call; nothing in those scripts changes.
if ~exist('GPISMAP_AMD', 'var')
    GPISMAP_AMD = fileparts(fileparts(mfilename('fullpath')));   % <checkout>/mex/.. 
end
INC = fullfile(GPISMAP_AMD, 'include');
LIB = fullfile(GPISMAP_AMD, 'gpismap_amd');
if ~exist('GATEWAY', 'var')
    GATEWAY = 'mexGPisMap3.cpp';          % the reference's file, in the current directory
end

disp('Running >> mex mexGPisMap3.cpp -lgpismap_amd (MI355X-native GPisMap3)...');
mex(GATEWAY, ...
    strcat('-I', INC), ...
    strcat('-L', LIB), '-lgpismap_amd', ...
    'CXXFLAGS=-std=c++11 -pthread -fPIC', ...
    ['LDFLAGS=$LDFLAGS -Wl,-rpath,' LIB], '-O');
