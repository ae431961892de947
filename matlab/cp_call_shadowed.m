function varargout = cp_call_shadowed(name, varargin)
% Calls the reference's OWN <name>.m -- the file that this directory shadows once it is on the path -- with the given
% arguments and the caller's number of outputs. Used by the same-named entry points here when they are handed numeric
% arguments or function handles (arbitrary closures cannot be fused into the engine): the reference's code then runs
% unchanged, so closure-based callers (the reference's test_* scripts, the drivers' F / G handles) keep working.
% The original is the first other <name>.m of which(name, '-all'); it is called with its own folder current, where
% MATLAB's precedence rules (current folder before path) resolve <name> to it. Nothing has to be renamed or copied.
persistent targets
if isempty(targets), targets = containers.Map(); end
if isKey(targets, name)
    target = targets(name);
else
    here = fileparts(mfilename('fullpath'));
    found = which(name, '-all');
    if ischar(found), found = {found}; end
    target = '';
    for i = 1:numel(found)
        if ~strcmp(fileparts(found{i}), here) && exist(found{i}, 'file') == 2
            target = found{i};
            break
        end
    end
    if isempty(target)
        error('cpengine:unsupported', ['%s was called with numeric or function-handle arguments: that runs the reference''s own ' ...
              '%s.m, but no other %s.m is on the MATLAB path. Add the reference''s folder that holds it (''Primal Dual ' ...
              'Optimization'' or ''Nuclear Norm Experiments'') to the path, below %s.'], name, name, name, here);
    end
    targets(name) = target;
end
old = cd(fileparts(target));
back = onCleanup(@() cd(old));
[varargout{1:nargout}] = feval(name, varargin{:});
end
